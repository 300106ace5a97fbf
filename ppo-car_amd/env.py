"""VecCarEnv -- the gymnasium vector-env surface train.py drives (reference train.py:138-142,
159-164,185,296), backed by the HIP env-step kernel through the C-ABI.  Tensors in, tensors out,
everything stays on the GPU; no host synchronisation in step()."""
import ctypes as C
import os

import numpy as np
import torch

from . import _capi
from ._capi import check, lib


class Track:
    """A loaded track (CarEnv.load_track, car_env.py:535-567).  Host-side only."""

    def __init__(self, path=None, *, walls=None, gates=None, start=None):
        h = C.c_void_p()
        if path is not None:
            # the reference prints "Track file not found" and returns None (car_env.py:624-628);
            # here a missing / malformed track is a hard error
            check(lib.pc_track_load_json(os.fsencode(path), C.byref(h)), f"pc_track_load_json({path!r})")
        else:
            w = np.ascontiguousarray(walls, np.float64).reshape(-1, 4)
            g = np.ascontiguousarray(gates, np.float64).reshape(-1, 4)
            check(lib.pc_track_from_arrays(w.ctypes.data, len(w), g.ctypes.data, len(g), float(start[0]), float(start[1]),
                                           float(start[2]), C.byref(h)), "pc_track_from_arrays")
        self._h = h
        self.path = path
        s, g_, st = C.c_int(), C.c_int(), (C.c_double * 3)()
        check(lib.pc_track_info(h, C.byref(s), C.byref(g_), st), "pc_track_info")
        self.n_walls, self.n_gates = s.value, g_.value
        self.start_x, self.start_y, self.start_angle = st[0], st[1], st[2]

    def geometry(self):
        walls = np.zeros((self.n_walls, 4), np.float64)
        gates = np.zeros((self.n_gates, 4), np.float64)
        check(lib.pc_track_geometry(self._h, walls.ctypes.data, gates.ctypes.data), "pc_track_geometry")
        return walls, gates

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.pc_track_destroy(h)


def ray_count(num_rays):
    """len(range(0, 360, 360 // n)) -- car_env.py:269: 12 -> 12, 16 -> 17, 32 -> 33."""
    r = lib.pc_ray_count(int(num_rays))
    check(min(r, 0), f"pc_ray_count({num_rays})")
    return r


class _Space:
    """gymnasium's two spaces as the reference declares and train.py reads them (train.py:141-142): the Box observation space
    with its bound vectors (car_env.py:513-524: low = [0, 0, -1, -1, -1, -1, 0 ...], high = 1, float32) and Discrete(9)
    (car_env.py:525).  `.shape` is the width CarEnv.step actually PRODUCES, 6 + R (R = 12 / 17 / 33 rays for num_rays 12 / 16 /
    32, car_env.py:269); the reference declares 6 + num_rays (its Box for num_rays = 16 says (22,) while _get_obs returns 23
    entries -- SURVEY quirk Q1): that declared shape is kept as `.declared_shape`."""

    def __init__(self, shape=None, n=None, dtype=np.float32, low=None, high=None, declared_shape=None):
        self.shape, self.n, self.dtype = shape, n, dtype
        self.low, self.high = low, high
        self.declared_shape = declared_shape if declared_shape is not None else shape

    @classmethod
    def car_obs(cls, obs_dim, num_rays_nominal, batch=None):
        """Box(low, high) of CarEnv.__init__ (car_env.py:514-524) for an observation of obs_dim entries; batch = n_envs gives the
        vector env's batched space (gymnasium tiles the single space's bounds)."""
        low = np.concatenate([np.array([0.0, 0.0, -1.0, -1.0, -1.0, -1.0], np.float32), np.zeros(obs_dim - 6, np.float32)])
        high = np.ones(obs_dim, np.float32)
        shape, decl = (obs_dim,), (6 + num_rays_nominal,)
        if batch is not None:
            low, high = np.broadcast_to(low, (batch, obs_dim)), np.broadcast_to(high, (batch, obs_dim))
            shape, decl = (batch, obs_dim), (batch, 6 + num_rays_nominal)
        return cls(shape=shape, low=low, high=high, declared_shape=decl)

    def contains(self, x):
        """gymnasium.spaces.Box.contains / Discrete.contains.  (The reference never clips the position to the track's frame, so a
        car that leaves the 1280 x 720 window produces an observation outside its own declared Box: car_env.py:578-581.)"""
        x = np.asarray(x)
        if self.n is not None:
            return bool(np.issubdtype(x.dtype, np.integer) and np.all((x >= 0) & (x < self.n)))
        if self.low is None:
            return tuple(x.shape) == tuple(self.shape)
        return bool(tuple(x.shape) == tuple(self.shape) and np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        if self.n is not None:
            return f"Discrete({self.n})"
        if self.low is None:
            return f"Box(-inf, inf, {self.shape}, float32)"
        return f"Box({float(np.min(self.low))}, {float(np.max(self.high))}, {self.shape}, float32)"


def _device_index(device):
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"VecCarEnv runs on an AMD GPU only (got device {device}); there is no CPU path")
    if not torch.cuda.is_available():
        raise RuntimeError("VecCarEnv needs a GPU: torch.cuda.is_available() is False and there is no CPU fallback")
    return device.index if device.index is not None else torch.cuda.current_device()


class VecCarEnv:
    """n_envs CarEnv instances stepped by one kernel launch.

    Mirrors `gym.vector.AsyncVectorEnv([make_env(...)] * n)` as the reference uses it:
      reset(options={"track_path": p}) -> (obs[N, D] float32, infos)          train.py:159-164
      step(actions[N] int64) -> (obs, rewards, terminateds, truncateds, infos)  train.py:185
        with TransformReward's `r * reward_scaling` (train.py:65,68) and same-step auto-reset
      close()                                                                   train.py:296
    `num_rays` is Car's nominal ray count (car_env.py:227); the observation has 6 + R entries with
    R = len(range(0, 360, 360 // num_rays)) exactly as the reference produces them.
    tracks: one path / Track, or a list of them with `track_id` [N] picking each env's track.
    """

    def __init__(self, n_envs, tracks, num_rays=12, reward_scaling=1.0, device="cuda", dtype="f32", track_id=None):
        self.device = torch.device("cuda", _device_index(device))
        self.num_envs = int(n_envs)
        self.num_rays = int(num_rays)
        self.reward_scaling = float(reward_scaling)
        self.dtype = dtype
        self._h = None
        self._tracks = None
        self._track_id = None if track_id is None else np.ascontiguousarray(track_id, np.uint8)
        self._opts = {}
        self._build(tracks)

    # ---- construction ---------------------------------------------------------------------
    def _build(self, tracks):
        if isinstance(tracks, (str, os.PathLike, Track)):
            tracks = [tracks]
        tr = [t if isinstance(t, Track) else Track(t) for t in tracks]
        arr = (C.c_void_p * len(tr))(*[t._h for t in tr])
        h = C.c_void_p()
        tid = self._track_id
        if tid is not None and len(tid) != self.num_envs:
            raise ValueError("track_id must have one entry per env")
        check(lib.pc_env_create(self.device.index, self.num_envs, self.num_rays, arr, len(tr),
                                tid.ctypes.data if tid is not None else None, _capi.DTYPES[self.dtype], C.byref(h)),
              "pc_env_create")
        self.close()
        self._h, self._tracks = h, tr
        self.obs_dim = lib.pc_env_obs_dim(h)
        self.act_dim = lib.pc_env_num_actions(h)
        # dtype f32: walls that float32 cannot order (crossing / touching non-neighbours, spikes, very short walls) are resolved by a
        # float64 scan of the whole chain -- exact, and an O(n_walls) cost on every ray that selects one: say so when it is a large share
        self.track_info = []
        for k in range(len(tr)):
            w, nv, ns = C.c_int(), C.c_int(), C.c_int()
            check(lib.pc_env_track_info(h, k, C.byref(w), C.byref(nv), C.byref(ns)), "pc_env_track_info")
            self.track_info.append({"n_walls": w.value, "n_chain_vertices": nv.value, "n_scan_segments": ns.value})
            if self.dtype in ("f32", "float32") and ns.value * 4 > w.value:
                import warnings
                warnings.warn(f"track {k} ({getattr(tr[k], 'path', None)}): {ns.value} of {w.value} walls cross, touch or fold back on another wall; "
                              "dtype='f32' resolves every ray that selects one of them by a float64 scan of all walls (exact, slow) -- "
                              "consider dtype='f64' or cleaning the track (python -m ppo_car_amd.track_tool check)", RuntimeWarning, stacklevel=3)
        # what train.py:141-142 reads: envs.single_observation_space.shape, envs.single_action_space.n
        self.single_observation_space = _Space.car_obs(self.obs_dim, self.num_rays)
        self.single_action_space = _Space(n=self.act_dim, dtype=np.int64)
        self.observation_space = _Space.car_obs(self.obs_dim, self.num_rays, batch=self.num_envs)
        self.action_space = _Space(shape=(self.num_envs,), dtype=np.int64)
        self.single_observation_space_shape = (self.obs_dim,)     # (round-1 spellings, kept)
        self.single_action_space_n = self.act_dim
        for name, value in self._opts.items():       # (a rebuilt handle -- reset(options={"track_path": ...}) -- keeps its options)
            check(lib.pc_env_set_option(self._h, self.OPTIONS[name], value), f"pc_env_set_option({name})")

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _new(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    @staticmethod
    def _ptr(t, dtype, numel, what):
        if t is None:
            return None
        if not (t.is_cuda and t.dtype == dtype and t.is_contiguous() and t.numel() == numel):
            raise ValueError(f"{what}: need a contiguous {dtype} CUDA tensor with {numel} elements, got "
                             f"{tuple(t.shape)} {t.dtype} {t.device}")
        return t.data_ptr()

    # ---- gymnasium-style surface ----------------------------------------------------------------
    def reset(self, seed=None, options=None, out=None):
        """seed is accepted and ignored, as in the reference (the env is deterministic, car_env.py:617)."""
        if options and "track_path" in options:
            if len(self._tracks) != 1 or self._tracks[0].path != options["track_path"]:
                self._build(options["track_path"])
        obs = out if out is not None else self._new(self.num_envs, self.obs_dim)
        check(lib.pc_env_reset(self._h, self._ptr(obs, torch.float32, self.num_envs * self.obs_dim, "obs"), self._stream()),
              "pc_env_reset")
        return obs, {}

    def infos(self):
        """CarEnv._get_info() (car_env.py:599-603) of every env's current state, as device tensors."""
        gp, tp = self._new(self.num_envs, dtype=torch.int32), self._new(self.num_envs, dtype=torch.int32)
        check(lib.pc_env_info(self._h, gp.data_ptr(), tp.data_ptr(), self._stream()), "pc_env_info")
        return {"gates_passed": gp, "time_passed": tp}

    def step(self, actions, out=None, gates_passed=None, final_obs=None, info=False):
        """out = (obs, rewards, terminateds, truncateds) preallocated tensors (e.g. rows of the rollout
        buffer) or None to allocate.  Flags are float32 0/1, what train.py:191-192 builds.
        info=True fills `infos` with CarEnv._get_info()'s keys (car_env.py:599-603) as device tensors:
        "gates_passed" / "time_passed" of every env's current state (0 / 0 for an env auto-reset in this step, whose
        finished episode's count is "final_gates_passed" -- gymnasium's final_info["gates_passed"])."""
        N, D = self.num_envs, self.obs_dim
        if actions.dtype != torch.int64 or not actions.is_cuda:
            actions = actions.to(device=self.device, dtype=torch.int64)
        actions = actions.contiguous()
        if out is None:
            out = (self._new(N, D), self._new(N), self._new(N), self._new(N))
        obs, rew, term, trunc = out
        if info and gates_passed is None:
            gates_passed = self._new(N, dtype=torch.int32)
        check(lib.pc_env_step(self._h, self._ptr(actions, torch.int64, N, "actions"), self.reward_scaling,
                              self._ptr(obs, torch.float32, N * D, "obs"), self._ptr(rew, torch.float32, N, "rewards"),
                              self._ptr(term, torch.float32, N, "terminateds"), self._ptr(trunc, torch.float32, N, "truncateds"),
                              self._ptr(gates_passed, torch.int32, N, "gates_passed"),
                              self._ptr(final_obs, torch.float32, N * D, "final_obs"), self._stream()), "pc_env_step")
        infos = {}
        if info:
            gp, tp = self._new(N, dtype=torch.int32), self._new(N, dtype=torch.int32)
            check(lib.pc_env_info(self._h, gp.data_ptr(), tp.data_ptr(), self._stream()), "pc_env_info")
            infos["gates_passed"], infos["time_passed"] = gp, tp
            infos["final_gates_passed"] = gates_passed
        elif gates_passed is not None:
            infos["gates_passed"] = gates_passed
        if final_obs is not None:
            infos["final_observation"] = final_obs
        return obs, rew, term, trunc, infos

    def step_many(self, actions, out=None):
        """`for t in range(T): envs.step(actions[t])` as ONE call (pc_env_step_many): actions [T, N] int64 -> (obs [T, N, D], rewards,
        terminateds, truncateds [T, N]), row t = what the t-th step() returns, bit for bit; the env state afterwards is the state after
        those T steps.  Where the handle has the table-driven kernel (last_step_kernel() == "K1f" / "K1f-table") the T steps are one launch."""
        N, D = self.num_envs, self.obs_dim
        if actions.dtype != torch.int64 or not actions.is_cuda:
            actions = actions.to(device=self.device, dtype=torch.int64)
        actions = actions.contiguous()
        if actions.dim() != 2 or actions.shape[1] != N:
            raise ValueError(f"step_many: actions must be [T, {N}], got {tuple(actions.shape)}")
        T = actions.shape[0]
        if out is None:
            out = (self._new(T, N, D), self._new(T, N), self._new(T, N), self._new(T, N))
        obs, rew, term, trunc = out
        check(lib.pc_env_step_many(self._h, self._ptr(actions, torch.int64, T * N, "actions"), T, self.reward_scaling,
                                   self._ptr(obs, torch.float32, T * N * D, "obs"), self._ptr(rew, torch.float32, T * N, "rewards"),
                                   self._ptr(term, torch.float32, T * N, "terminateds"), self._ptr(trunc, torch.float32, T * N, "truncateds"),
                                   self._stream()), "pc_env_step_many")
        return obs, rew, term, trunc

    def last_step_kernel(self):
        """which kernel the last step() / step_many() launched: "K1" (generic per-step kernel), "K1f" (table-driven), "K1f-table" or "none"."""
        code = lib.pc_env_last_step_kernel(self._h)
        if code < 0:
            check(code, "pc_env_last_step_kernel")
        return _capi.PC_STEP_NAMES[code]

    def close(self):
        h, self._h = self._h, None
        if h:
            lib.pc_env_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- teacher forcing / introspection (parity tests, bench) -----------------------------------
    _FIELDS = (("px", np.float64), ("py", np.float64), ("vx", np.float64), ("vy", np.float64), ("rot", np.float64),
               ("time_step", np.int64), ("next_gate", np.int64), ("passed", np.int64))

    def get_state(self):
        out = {k: np.zeros(self.num_envs, dt) for k, dt in self._FIELDS}
        check(lib.pc_env_get_state(self._h, *[out[k].ctypes.data for k, _ in self._FIELDS]), "pc_env_get_state")
        return out

    def set_state(self, **kw):
        args = []
        for k, dt in self._FIELDS:
            v = kw.pop(k, None)
            if v is None:
                args.append(None)
            else:
                a = np.ascontiguousarray(np.broadcast_to(np.asarray(v, dt), (self.num_envs,)))
                args.append(a)
        if kw:
            raise TypeError(f"unknown state fields {sorted(kw)}")
        check(lib.pc_env_set_state(self._h, *[a.ctypes.data if a is not None else None for a in args]), "pc_env_set_state")

    def launch_info(self):
        v = [C.c_int() for _ in range(4)]
        check(lib.pc_env_launch_info(self._h, *[C.byref(x) for x in v]), "pc_env_launch_info")
        return dict(lanes_per_env=v[0].value, rays_per_lane=v[1].value, blocks=v[2].value, threads=v[3].value)

    def last_rollout_kernel(self):
        """which persistent kernel the last pc_rollout on this handle launched: "K9", "K9s", "K9-literal", "K9d-filter" or "none" (every one fills the same buffers bit for bit)"""
        from ._capi import PC_KERNEL_NAMES
        code = lib.pc_env_last_rollout_kernel(self._h)
        if code < 0:
            check(code, "pc_env_last_rollout_kernel")
        return PC_KERNEL_NAMES[code]

    def set_lanes_per_env(self, lanes):
        check(lib.pc_env_set_lanes_per_env(self._h, int(lanes)), "pc_env_set_lanes_per_env")

    OPTIONS = {"rollout_form": 1, "rollout_epw": 2, "rollout_fast": 3, "step_form": 4}     # PC_OPT_* (include/ppocar.h)

    def set_option(self, name, value):
        """Per-handle launch option of pc_rollout / pc_env_step on THIS vector env (other handles keep theirs).  step_form: 0 automatic,
        1 always the generic per-step kernel K1, 2 the table-driven K1f wherever the handle has it."""
        check(lib.pc_env_set_option(self._h, self.OPTIONS[name], int(value)), f"pc_env_set_option({name})")
        self._opts[name] = int(value)

    def get_option(self, name):
        v = C.c_int()
        check(lib.pc_env_get_option(self._h, self.OPTIONS[name], C.byref(v)), f"pc_env_get_option({name})")
        return v.value
