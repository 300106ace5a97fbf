"""ctypes binding of libppocar.so (include/ppocar.h).  The library is the product; there is no
Python or CPU fallback: if it is missing, importing this module fails loudly."""
import ctypes as C
import os

# torch FIRST: PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Loading it before
# libppocar.so makes the dynamic linker bind our library to that same HIP runtime, so torch's streams,
# events and device pointers are valid inside our launches.  The other order would put two HIP runtimes
# in one process.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libppocar.so")

PC_OK = 0
PC_ERR_INVALID_ARG, PC_ERR_IO, PC_ERR_PARSE, PC_ERR_HIP, PC_ERR_UNSUPPORTED, PC_ERR_NO_DEVICE, PC_ERR_TIMEOUT = -1, -2, -3, -4, -5, -6, -7
PC_XCHG_HANDLE_BYTES = 128
PC_DTYPE_F32, PC_DTYPE_F64 = 0, 1
PC_OPT_ROLLOUT_FORM, PC_OPT_ROLLOUT_EPW, PC_OPT_ROLLOUT_FAST = 1, 2, 3
PC_KERNEL_NAMES = {0: "none", 1: "K9", 2: "K9s", 3: "K9-literal", 4: "K9d-filter", 5: "K9s-literal", 6: "K9d-selector", 7: "K9m", 8: "K9m-literal"}     # pc_env_last_rollout_kernel
PC_STEP_NAMES = {0: "none", 1: "K1", 2: "K1f", 3: "K1f-table"}     # pc_env_last_step_kernel
DTYPES = {"f32": PC_DTYPE_F32, "float32": PC_DTYPE_F32, "f64": PC_DTYPE_F64, "float64": PC_DTYPE_F64}


class PpoCarError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        msg = f"{what}: {lib.pc_strerror(code).decode()} (code {code})"
        detail = lib.pc_last_hip_error().decode()
        if detail and code in (PC_ERR_HIP, PC_ERR_UNSUPPORTED):
            msg += f" [{detail}]"
        super().__init__(msg)


def lib_path():
    return _LIB_PATH


if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"{_LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C ppo-car_amd/csrc`). There is no CPU fallback for the CarEnv hot path.")

lib = C.CDLL(_LIB_PATH)


def _hip_runtimes():
    try:
        with open("/proc/self/maps") as f:
            return sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    except OSError:
        return []


if len(_hip_runtimes()) > 1:
    raise ImportError(f"two HIP runtimes are mapped in this process ({_hip_runtimes()}): libppocar.so must share "
                      "PyTorch's libamdhip64 -- import torch before anything that loads /opt/rocm's copy")

_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_sig = {
    "pc_track_load_json": (_i, [C.c_char_p, C.POINTER(_vp)]),
    "pc_track_from_arrays": (_i, [_vp, _i, _vp, _i, _d, _d, _d, C.POINTER(_vp)]),
    "pc_track_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_d)]),
    "pc_track_geometry": (_i, [_vp, _vp, _vp]),
    "pc_track_destroy": (None, [_vp]),
    "pc_ray_count": (_i, [_i]),
    "pc_env_create": (_i, [_i, _i64, _i, C.POINTER(_vp), _i, _vp, _i, C.POINTER(_vp)]),
    "pc_env_destroy": (None, [_vp]),
    "pc_env_obs_dim": (_i, [_vp]),
    "pc_env_num_actions": (_i, [_vp]),
    "pc_env_num_envs": (_i64, [_vp]),
    "pc_env_reset": (_i, [_vp, _vp, _vp]),
    "pc_env_step": (_i, [_vp, _vp, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pc_env_step_many": (_i, [_vp, _vp, C.c_int64, _d, _vp, _vp, _vp, _vp, _vp]),
    "pc_env_last_step_kernel": (_i, [_vp]),
    "pc_env_info": (_i, [_vp, _vp, _vp, _vp]),
    "pc_build_ablate": (_i, []),
    "pc_env_get_state": (_i, [_vp] * 9),
    "pc_env_set_state": (_i, [_vp] * 9),
    "pc_gae": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _i64, _i64, _vp, _vp, _vp]),
    "pc_sample": (_i, [_i, _vp, _i64, _i, C.c_uint64, C.c_uint64, _vp, _vp, _vp, _vp]),
    "pc_policy_create": (_i, [_i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "pc_policy_destroy": (None, [_vp]),
    "pc_policy_get": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i64)]),
    "pc_policy_pack": (_i, [_vp] + [_vp] * 8 + [_vp, _vp]),
    "pc_policy_pack_checked": (_i, [_vp] + [_vp] * 8 + [_vp, _vp, _vp]),
    "pc_policy_act": (_i, [_vp, _vp, _i64, _vp, C.c_uint64, C.c_uint64, _vp] + [_vp] * 5 + [_vp]),
    "pc_rollout": (_i, [_vp, _vp, _vp, _i64, _d, C.c_uint64, C.c_uint64, _vp] + [_vp] * 12 + [_vp]),
    "pc_env_set_option": (_i, [_vp, _i, _i]),
    "pc_env_get_option": (_i, [_vp, _i, C.POINTER(_i)]),
    "pc_ppo_gather": (_i, [_i, _vp, _i, _i] + [_vp] * 10 + [_vp]),
    "pc_ppo_loss": (_i, [_i] + [_vp] * 6 + [_i, _i, _d, _d, _d, _vp, _vp, _vp, _vp]),
    "pc_clip_adam": (_i, [_i] + [_vp] * 6 + [_i64, _d, _d, _d, _d, _d, _vp]),
    "pc_clip_adam_advanced": (_i, [_i] + [_vp] * 6 + [_i64, _d, _d, _d, _d, _d, _vp]),
    "pc_ppo_workspace_floats": (_i64, [_i, _i, _i, _i]),
    "pc_ppo_prepared_floats": (_i64, [_i, _i]),
    "pc_ppo_prepare": (_i, [_i, _vp, _i64, _i, _i, _i] + [_vp] * 5 + [_vp, _vp]),
    "pc_ppo_minibatch_prepared": (_i, [_i, _vp, _i, _i, _i, _i] + [_vp] * 6 + [_d] * 7 + [_vp, _vp, _i, _vp]),
    "pc_ppo_epoch_state_floats": (_i64, [_i, _i, _i]),
    "pc_ppo_epoch_prepared": (_i, [_i, _vp, _i, _i, _i, _i, _i] + [_vp] * 6 + [_d] * 7 + [_vp, _vp, _vp, _vp]),
    "pc_ppo_minibatch": (_i, [_i, _vp, _i, _i, _i, _i] + [_vp] * 5 + [_vp] * 6 + [_d] * 7 + [_vp, _vp, _i, _vp]),
    "pc_xchg_create": (_i, [_i, _i, _i, _i64, C.POINTER(_vp)]),
    "pc_xchg_local_handle": (_i, [_vp, _vp]),
    "pc_xchg_connect": (_i, [_vp, _vp]),
    "pc_xchg_connect_local": (_i, [_vp, _vp]),
    "pc_xchg_allreduce_group": (_i, [_vp, _vp, _vp]),
    "pc_xchg_set_timeout": (_i, [_vp, _d]),
    "pc_xchg_allreduce": (_i, [_vp, _vp, _vp]),
    "pc_xchg_status": (_i, [_vp]),
    "pc_xchg_destroy": (None, [_vp]),
    "pc_strerror": (C.c_char_p, [_i]),
    "pc_last_hip_error": (C.c_char_p, []),
    "pc_env_launch_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "pc_env_set_lanes_per_env": (_i, [_vp, _i]),
    "pc_env_track_info": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "pc_env_last_rollout_kernel": (_i, [_vp]),
}
for _name, (_res, _args) in _sig.items():
    _f = getattr(lib, _name)  # AttributeError here = the library does not export what ppocar.h declares
    _f.restype, _f.argtypes = _res, _args

EXPORTS = tuple(_sig)

if lib.pc_build_ablate() != 0:
    raise ImportError(f"{_LIB_PATH} is a developer timing-ablation build (PC_ABLATE={lib.pc_build_ablate()}): "
                      "its rollout kernel skips work. Rebuild the product library with `make -C ppo-car_amd/csrc`.")


def check(code, what):
    if code != PC_OK:
        raise PpoCarError(code, what)
