"""CPU-side tests of the product library: it loads, exports every symbol include/ppocar.h declares,
its host logic (track loader, ray count, error codes) matches the reference, and the compute entry
points refuse to run without a GPU instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import ppo_car_amd as pc
from ppo_car_amd import _capi
from conftest import ENV_CONFIGS, GOLDEN, ROOT, TRACKS


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ppocar.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = C.CDLL(pc.lib_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in ppocar.h but not exported"
    assert declared == set(_capi.EXPORTS)


def test_library_is_in_tree():
    assert os.path.dirname(pc.lib_path()) == os.path.join(ROOT, "ppo-car_amd")


@pytest.mark.parametrize("track,n", ENV_CONFIGS[::3])
def test_track_loader_matches_reference(track, n):
    g = np.load(f"{GOLDEN}/env_{track}_n{n}.npz")
    t = pc.Track(TRACKS[track])
    walls, gates = t.geometry()
    assert np.array_equal(walls, g["walls"]) and np.array_equal(gates, g["gates"])   # bit-exact float64
    assert (t.n_walls, t.n_gates) == {"big_track": (24, 55), "track": (16, 45), "oval64": (128, 40)}[track]
    assert np.array_equal([t.start_x, t.start_y, t.start_angle], g["reset_state"][[0, 1, 4]])


def test_track_errors(tmp_path):
    with pytest.raises(pc.PpoCarError) as e:
        pc.Track(str(tmp_path / "nope.json"))
    assert e.value.code == _capi.PC_ERR_IO     # reference prints + returns None (car_env.py:626-628); here: hard error
    bad = tmp_path / "bad.json"
    bad.write_text('{"outer_track_points": [[0.1, 0.2], [0.3')
    with pytest.raises(pc.PpoCarError) as e:
        pc.Track(str(bad))
    assert e.value.code == _capi.PC_ERR_PARSE
    missing_key = tmp_path / "nokey.json"
    missing_key.write_text('{"outer_track_points": [[0,0],[1,1]], "inner_track_points": [[0,0],[1,1]], "reward_gates": [[0,0],[1,1]]}')
    with pytest.raises(pc.PpoCarError) as e:
        pc.Track(str(missing_key))
    assert e.value.code == _capi.PC_ERR_PARSE
    wrong_shape = tmp_path / "shape.json"
    wrong_shape.write_text('{"outer_track_points": [[0,0,1],[1,1]], "inner_track_points": [[0,0],[1,1]], '
                           '"reward_gates": [[0,0],[1,1]], "initial_position": [0.5, 0.5], "initial_angle": 0}')
    with pytest.raises(pc.PpoCarError):
        pc.Track(str(wrong_shape))


def test_track_json_variants(tmp_path):
    """whitespace, exponents, negative numbers, extra keys, escaped strings"""
    p = tmp_path / "t.json"
    p.write_text('{\n "name": "a \\"quoted\\" one", "outer_track_points": [[1e-1, 2.5E-1], [ 0.3 ,0.4],[0.5,0.6]],\n'
                 '"inner_track_points": [[0.2,0.2],[0.25,0.3]], "reward_gates": [[0.1,0.1],[0.2,0.2],[0.3,0.3],[0.4,0.4],[0.9,0.9]],'
                 '"initial_position": [0.5, 0.25], "initial_angle": -12.5, "extra": {"a": [true, false, null]}}')
    t = pc.Track(str(p))
    walls, gates = t.geometry()
    assert t.n_walls == 3 and t.n_gates == 2          # trailing unpaired gate point is dropped, like zip()
    assert np.array_equal(walls[0], [0.1 * 1280, 0.25 * 720, 0.3 * 1280, 0.4 * 720])
    assert np.array_equal(walls[2], [0.2 * 1280, 0.2 * 720, 0.25 * 1280, 0.3 * 720])
    assert np.array_equal(gates[1], [0.3 * 1280, 0.3 * 720, 0.4 * 1280, 0.4 * 720])
    assert (t.start_x, t.start_y, t.start_angle) == (0.5 * 1280, 0.25 * 720, -12.5)


@pytest.mark.parametrize("n", [4, 8, 12, 16, 24, 32, 36, 45, 72, 90, 360])
def test_ray_count(n):
    from ppo_car_amd.env import ray_count
    assert ray_count(n) == len(range(0, 360, 360 // n))


def test_ray_count_invalid():
    from ppo_car_amd.env import ray_count
    for n in (0, 3, -5, 361):
        with pytest.raises(pc.PpoCarError):
            ray_count(n)


def test_track_from_arrays_roundtrip():
    w = np.arange(12, dtype=np.float64).reshape(3, 4)
    g = np.arange(8, dtype=np.float64).reshape(2, 4) + 100
    t = pc.Track(walls=w, gates=g, start=(1.0, 2.0, 3.0))
    ww, gg = t.geometry()
    assert np.array_equal(ww, w) and np.array_equal(gg, g) and (t.start_x, t.start_y, t.start_angle) == (1.0, 2.0, 3.0)


def test_strerror():
    assert _capi.lib.pc_strerror(0) == b"ok"
    for c in range(-6, 0):
        assert len(_capi.lib.pc_strerror(c)) > 3


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    with pytest.raises(RuntimeError, match="no CPU"):
        pc.VecCarEnv(4, TRACKS["big_track"])
    with pytest.raises(RuntimeError, match="GPU only"):
        pc.VecCarEnv(4, TRACKS["big_track"], device="cpu")
    buf = pc.Buffer((18,), 2, 3, torch.device("cpu"))
    for _ in range(2):
        buf.store(torch.zeros(3, 18), torch.zeros(3), torch.zeros(3), torch.zeros(3), torch.zeros(3), torch.zeros(3), torch.zeros(3))
    with pytest.raises(RuntimeError, match="HIP GAE kernel"):
        buf.calculate_advantages(torch.zeros(1, 3), torch.zeros(1, 3), torch.zeros(1, 3))
    # the C-ABI itself reports "no device" rather than computing on the host
    t = pc.Track(TRACKS["big_track"])
    arr = (C.c_void_p * 1)(t._h)
    h = C.c_void_p()
    rc = _capi.lib.pc_env_create(0, 8, 12, arr, 1, None, 0, C.byref(h))
    assert rc in (_capi.PC_ERR_NO_DEVICE, _capi.PC_ERR_HIP)


def test_product_never_touches_the_oracle():
    """The product tree must not import / link / call anything under oracle/."""
    pkg = os.path.join(ROOT, "ppo-car_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(import|from)\s+oracle", src, flags=re.M), f
                assert "liboracle" not in src and "carenv_oracle" not in src, f
    # every top-level script and example of the product, the import shim, the C header
    scripts = ["train.py", "evaluate.py", os.path.join("ppo_car_amd", "__init__.py")]
    scripts += [os.path.join("examples", f) for f in sorted(os.listdir(os.path.join(ROOT, "examples"))) if f.endswith(".py")]
    scripts += [os.path.join("include", f) for f in sorted(os.listdir(os.path.join(ROOT, "include")))]
    assert "evaluate.py" in scripts and any(s.startswith("examples") for s in scripts)
    for f in scripts:
        src = open(os.path.join(ROOT, f)).read()
        assert not re.search(r"^\s*(import|from)\s+oracle", src, flags=re.M), f
        assert "liboracle" not in src and "carenv_oracle" not in src, f


def test_bench_uses_the_oracle_only_as_checker_and_cpu_baseline():
    """bench.py may touch oracle/ in exactly two places: the `cpu_baseline` leg and the `parity_check` legs (parity_check and its second,
    rare-branch leg parity_check_rare; all run AFTER the timed region, on rank 0).  Nothing else in the file -- the timed region in main() least of all -- names it."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    allowed = {"cpu_baseline", "parity_check", "parity_check_rare"}
    users = set()

    def walk(node, fn):
        for ch in ast.iter_child_nodes(node):
            name = ch.name if (isinstance(ch, (ast.FunctionDef, ast.AsyncFunctionDef)) and fn == "<module>") else fn   # the TOP-LEVEL function
            if isinstance(ch, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in ch.names):
                users.add(fn)
            if isinstance(ch, ast.ImportFrom) and (ch.module or "").split(".")[0] == "oracle":
                users.add(fn)
            if isinstance(ch, ast.Name) and ch.id == "oracle":
                users.add(fn)
            walk(ch, name)

    walk(tree, "<module>")
    assert users and users <= allowed, users
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    assert "oracle" not in {n.id for n in ast.walk(main) if isinstance(n, ast.Name)}
    # ... and main() calls the two legs only after the timed region (`dt = timed(tr, args.steps)`)
    timed_line = next(n.lineno for n in ast.walk(main) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "timed")
    for n in ast.walk(main):
        if isinstance(n, ast.Call) and getattr(n.func, "id", "") in allowed:
            assert n.lineno > timed_line, (n.func.id, n.lineno)


def test_buffer_store_get_semantics_cpu_storage():
    """store()/get() are plain tensor plumbing (buffer.py:22-34,66-73) and work on any device."""
    T, N, D = 3, 4, 5
    buf = pc.Buffer((D,), T, N, torch.device("cpu"))
    for t in range(T):
        buf.store(torch.full((N, D), float(t)), torch.full((N,), 2.0), torch.full((N,), 0.5), torch.zeros(N), torch.zeros(N),
                  torch.zeros(N), torch.full((N,), -1.0))
    assert buf.ptr == T
    obs, act, val, lp = buf.get()
    assert buf.ptr == 0 and obs.shape == (T, N, D) and act.dtype == torch.float32
    assert torch.equal(obs[2], torch.full((N, D), 2.0)) and torch.equal(lp, torch.full((T, N), -1.0))
    with pytest.raises(AssertionError):
        buf.get()


def test_shape_menus_are_reported_without_a_gpu():
    """The size / precision queries validate shapes on the host: callers use them to decide on their fallbacks."""
    lib = _capi.lib
    import ctypes as C

    def form(D, H, A, precision=-1, split=-1):
        """(status, precision the shape got, split, image floats) of a pc_policy handle -- creation needs no GPU"""
        h = C.c_void_p()
        rc = lib.pc_policy_create(0, D, H, A, precision, split, C.byref(h))
        if rc != 0:
            return rc, None, None, None
        pr, sp, n = C.c_int(), C.c_int(), C.c_int64()
        assert lib.pc_policy_get(h, C.byref(pr), C.byref(sp), C.byref(n)) == 0
        lib.pc_policy_destroy(h)
        return rc, pr.value, sp.value, n.value

    assert form(23, 256, 9)[3] > 0 and form(39, 256, 9)[3] > 0
    assert form(23, 128, 9)[0] == _capi.PC_ERR_UNSUPPORTED      # hidden size other than 256
    assert form(41, 256, 9)[0] == _capi.PC_ERR_UNSUPPORTED
    assert form(23, 256, 16)[0] == _capi.PC_ERR_UNSUPPORTED
    assert form(23, 256, 9)[1] == 2 and form(39, 256, 9)[1] == 2    # the default form is fp16x2; the split forms cover D <= 40
    assert form(23, 256, 12)[1] == 0                                 # ... and A <= 9
    assert form(23, 256, 9, precision=0)[1] == 0 and form(23, 256, 9, precision=1)[1] == 1
    assert form(23, 256, 9, precision=7)[0] == _capi.PC_ERR_INVALID_ARG
    assert form(23, 256, 9)[3] == 32 * 2 * 48 * 4 + 8 * 2 * 4 * 10 * 4 + 512 + 16 + 256
    assert form(39, 256, 9)[3] == 32 * 2 * 80 * 4 + 8 * 2 * 4 * 10 * 4 + 512 + 16 + 256   # two K blocks, 5 stored groups
    assert form(23, 256, 9, split=2)[0] == _capi.PC_ERR_INVALID_ARG and form(23, 256, 9, split=-2)[0] == _capi.PC_ERR_INVALID_ARG
    assert form(23, 256, 9, split=1)[2] == 1
    n = lib.pc_ppo_workspace_floats(512, 23, 256, 9)
    n_param = 2 * (256 * 23 + 256) + 9 * 256 + 9 + 256 + 1
    n_pad = (n_param + 3) // 4 * 4                      # a partial's row stride: 16-byte aligned rows
    assert n_param == 14858 and n == 64 * n_pad + 64 * 4 + (n_param + 255) // 256
    assert lib.pc_ppo_workspace_floats(2048, 23, 256, 9) == _capi.PC_ERR_UNSUPPORTED    # batch above 1024
    assert lib.pc_ppo_workspace_floats(512, 23, 64, 9) == _capi.PC_ERR_UNSUPPORTED


def test_bench_algorithmic_counts_are_surveys_figures_for_every_ray_count_and_track():
    """bench.py prices a launch with SURVEY 8(d)'s expressions evaluated for the ACTUAL ray count and each track's wall count: they
    reproduce the survey's table (156 / 176 / 240 B; 8.2 / 11.6 / 22.4 kflop on big_track's 24 walls; 5.6 / 7.8 / 15.0 kflop on
    track.json's 16) and never fall back to zero flops for another ray count; a mixed batch is the mean over its tracks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert [b.step_bytes(R) for R in (12, 17, 33)] == [156, 176, 240]
    for R, S, kflop in ((12, 24, 8.2), (17, 24, 11.6), (33, 24, 22.4), (12, 16, 5.6), (17, 16, 7.8), (33, 16, 15.0)):
        assert abs(b.step_flops(R, [S]) / 1e3 - kflop) < 0.06, (R, S)
    assert b.step_flops(19, [24]) > b.step_flops(17, [24]) > 0                      # (18 -> 19 rays: priced, not zero)
    assert b.step_flops(17, [16, 24]) == (b.step_flops(17, [16]) + b.step_flops(17, [24])) / 2
    host, usable, quota = b.usable_cpus()
    assert 1 <= usable <= host and (quota is None or quota > 0)


def test_bench_main_never_reads_a_name_after_deleting_it_or_shadows_a_module_function():
    """bench.py's main() is 400 lines that only run end to end on a GPU box: two slips of the kind a CPU suite can still catch --
    a local read after its `del` (the trainer is deleted before the side measurements), a local that shadows a module-level
    function it later calls."""
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    top = {n.name for n in tree.body if isinstance(n, ast.FunctionDef)}
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    own = lambda f: [n for n in ast.walk(f)]
    deleted = {}
    for n in own(main):
        if isinstance(n, ast.Delete):
            for tg in n.targets:
                if isinstance(tg, ast.Name):
                    deleted[tg.id] = max(deleted.get(tg.id, 0), n.lineno)
    stores = {n.id for n in own(main) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store)}
    late = [(n.id, n.lineno) for n in own(main) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id in deleted and n.lineno > deleted[n.id]]
    # (names assigned again after their del -- tr2, tr3 inside their own blocks -- are stores, not loads; a load after the last del is the slip)
    assert not late, late
    assert not (stores & top), stores & top


def test_bench_refuses_more_gpus_than_are_visible_without_starting_ranks():
    """`python bench.py --gpus N` where fewer than N devices are visible (here: none) and no --same-device: the PARENT exits with code 2
    and a one-line reason on stderr before it starts any rank -- the first real multi-GPU run must not die inside a rendezvous for this."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "--gpus 8 but only" in r.stderr and "--same-device" in r.stderr
