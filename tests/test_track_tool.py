"""Track JSON authoring / validation tool: schema and geometry checks on the shipped tracks and on
broken variants, normalisation as track_editor.py does it, and generated many-segment circuits -- which must load
through the C-ABI and, on the GPU, step exactly like the oracle (walls >> the 24 of big_track: the 1/den table no longer
fits LDS there and the persistent kernel takes its arithmetic path)."""
import copy
import json

import numpy as np
import pytest
import torch

import oracle
import ppo_car_amd as pc
from conftest import TRACKS
from ppo_car_amd import track_tool as tt


def _load(name):
    return json.load(open(TRACKS[name]))


@pytest.mark.parametrize("name", ["big_track", "track"])
def test_shipped_tracks_validate_and_agree_with_the_loader(name):
    d = _load(name)
    assert tt.check_schema(d) == [] and tt.check_geometry(d) == []
    info = tt.summary(d, TRACKS[name])                      # raises if the C-ABI loader counts differ
    assert info["walls"] == len(d["outer_track_points"]) + len(d["inner_track_points"]) - 2
    assert info["gates"] == len(d["reward_gates"]) // 2 and info["chain_vertices_padded"] % 4 == 0
    assert tt.main(["validate", TRACKS[name]]) == 0


def test_broken_tracks_are_flagged():
    d = _load("big_track")
    bad = copy.deepcopy(d); bad["outer_track_points"].pop()             # loop not closed
    assert any("not closed" in p for p in tt.check_schema(bad))
    bad = copy.deepcopy(d); bad["reward_gates"].pop()                   # odd number of gate points
    assert any("odd number" in p for p in tt.check_schema(bad))
    bad = copy.deepcopy(d); del bad["initial_angle"]
    assert tt.check_schema(bad) == ["missing key 'initial_angle'"]
    bad = copy.deepcopy(d); bad["inner_track_points"][1] = [1.5, 0.2]   # outside the window
    assert any("outside the normalised window" in p for p in tt.check_schema(bad))
    bad = copy.deepcopy(d); bad["outer_track_points"][2] = [0.123456, 0.5]
    assert any("more than 4 decimals" in p for p in tt.check_schema(bad))
    bad = copy.deepcopy(d); bad["initial_position"] = [0.01, 0.01]      # outside the outer wall
    assert tt.check_schema(bad) == [] and any("initial_position" in p for p in tt.check_geometry(bad))
    bad = copy.deepcopy(d); bad["reward_gates"][0] = bad["reward_gates"][1]   # a gate of zero length beside one wall
    assert any("gate 0" in p for p in tt.check_geometry(bad))


def test_normalise_closes_loops_and_rounds_like_the_editor(tmp_path):
    d = _load("track")
    raw = copy.deepcopy(d)
    raw["outer_track_points"] = [[x + 1e-6, y - 1e-6] for x, y in d["outer_track_points"][:-1]]     # unrounded, open
    raw["inner_track_points"] = d["inner_track_points"][:-1] + [d["inner_track_points"][-2]] * 2    # duplicates, open
    n = tt.normalise(raw)
    assert tt.check_schema(n) == [] and n["outer_track_points"] == d["outer_track_points"]
    assert n["inner_track_points"] == d["inner_track_points"]
    src, dst = tmp_path / "raw.json", tmp_path / "norm.json"
    json.dump(raw, open(src, "w"))
    assert tt.main(["normalise", str(src), str(dst)]) == 0 and json.load(open(dst)) == n


@pytest.mark.parametrize("points,gates", [(16, 12), (64, 40), (200, 90)])
def test_generated_circuits_validate_and_load(tmp_path, points, gates):
    path = str(tmp_path / "oval.json")
    assert tt.main(["make-oval", path, "--points", str(points), "--gates", str(gates), "--wobble", "0.01"]) == 0
    d = json.load(open(path))
    assert tt.check_schema(d) == [] and tt.check_geometry(d) == []
    info = tt.summary(d, path)
    assert info["walls"] == 2 * points and info["loader"]["gates"] == gates
    t = oracle.Track(path)                                   # the oracle reads the same file
    assert (t.S, t.G) == (2 * points, gates)


@pytest.mark.gpu
# (24 points: two loops of 25 chain vertices = 13 groups of four over the small form's 8 sweep parts -- one or two groups per lane
# group of the wave-owned env step, and no room for the 1/den table beside the weight image: its arithmetic variant)
@pytest.mark.parametrize("points,gates,n", [(24, 20, 16), (64, 40, 16), (200, 90, 12)])
def test_generated_circuit_steps_like_the_oracle_and_through_the_persistent_kernel(tmp_path, points, gates, n):
    from ppo_car_amd.ppo import PPOConfig, Trainer
    path = str(tmp_path / "oval.json")
    assert tt.main(["make-oval", path, "--points", str(points), "--gates", str(gates), "--wobble", "0.01"]) == 0
    T, N = 300, 128
    rng = np.random.default_rng(points)
    a = rng.integers(0, 9, size=(T, N))
    fwd = rng.random((T, N)) < 0.5
    a[fwd] = rng.choice([0, 4, 5], size=int(fwd.sum()))
    a = a.astype(np.int64)
    oenv = oracle.OracleVecEnv(oracle.Track(path), N, num_rays=n, reward_scaling=0.1, threads=8)
    acts = torch.from_numpy(a).cuda()
    for dtype in ("f64", "f32"):
        env = pc.VecCarEnv(N, path, num_rays=n, reward_scaling=0.1, dtype=dtype)
        env.reset()
        oenv.reset()
        worst, n_gate, n_term = 0.0, 0, 0
        alive = np.ones(N, bool)                             # f32: env still on the oracle's trajectory (no near-tie flipped)
        for t in range(T):
            O, R, TE, TR = oenv.step(a[t])
            obs, rew, term, trunc, _ = env.step(acts[t])
            o, r, te, tr_ = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy() != 0, trunc.cpu().numpy() != 0
            if dtype == "f64":                               # the reference's own arithmetic: bit-exact
                assert np.array_equal(o, O) and np.array_equal(r, R.astype(np.float32))
                assert np.array_equal(te, TE) and np.array_equal(tr_, TR)
                n_gate += int((R > 0.09).sum()); n_term += int(TE.sum())
            else:                                            # float32 ray geometry: free-running, compared until a threshold near-tie flips
                alive &= ~((te != TE) | (tr_ != TR) | (r != R.astype(np.float32)))
                worst = max(worst, float(np.abs(o - O)[alive].max()) if alive.any() else 0.0)
        if dtype == "f64":
            assert n_term > 20 and n_gate > 20               # walls were hit and gates passed on the generated circuit
        else:
            assert worst <= 1e-5, worst                      # observations within the north-star tolerance
            assert alive.mean() > 0.95, alive.mean()         # and (nearly) every env's events identical over 300 steps
        env.close()
    # the persistent rollout kernel on this track (1/den table too large for LDS above ~90 walls: arithmetic path)
    # ... in both dtypes.  float64: 48 walls fit the selector's LDS tables (the small form's literal kernel, without the 1/den table);
    # the larger circuits take the generic kernel K9d with the per-step kernel's selector step, a big-form kernel (the unsplit policy arithmetic)
    for dtype in ("f32", "f64"):
        small_lit = dtype == "f64" and points == 24
        res = {}
        for mode in ("steps", "mega"):
            cfg = PPOConfig(n_envs=512, n_steps=64, num_rays=n, track=path, rollout_kernel=mode, use_graphs=False, seed=3, env_dtype=dtype,
                            policy_split=-1 if (dtype == "f32" or small_lit) else 0)
            tr = Trainer(cfg, device="cuda")
            tr.rollout()
            torch.cuda.synchronize()
            assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager")
            if mode == "mega" and dtype == "f64":
                assert tr.envs.last_rollout_kernel() == ("K9s-literal" if small_lit else "K9d-selector")
            b = tr.buffer
            res[mode] = [x.clone() for x in (b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.term_buf, tr.next_obs)]
            tr.close()
        for x, y in zip(res["steps"], res["mega"]):
            assert torch.equal(x, y), dtype


def test_rasteriser_draws_track_car_and_rays(tmp_path):
    """ppo_car_amd.render: walls, gates, rays and the car land where the geometry says; the PNG
    writer produces a file zlib / struct can read back."""
    import struct, zlib
    from ppo_car_amd.env import Track
    from ppo_car_amd.render import COLORS, rasterise, write_png
    t = Track(TRACKS["big_track"])
    walls, gates = t.geometry()
    rays = np.full(12, 0.05, np.float32)                       # 50 px in every direction
    img = rasterise(walls, gates, t.start_x, t.start_y, 30.0, rays, next_gate=3, num_rays_nominal=12)
    assert img.shape == (360, 640, 3) and img.dtype == np.uint8
    def has(color):
        return bool((img == np.array(color, np.uint8)).all(-1).any())
    assert all(has(COLORS[k]) for k in ("wall", "gate", "next_gate", "ray", "car", "background"))
    x, y = int(round(walls[0][0] / 2)), int(round(walls[0][1] / 2))        # a wall endpoint, at half scale
    assert (img[y, x] == np.array(COLORS["wall"], np.uint8)).all()
    path = tmp_path / "f.png"
    write_png(str(path), img)
    b = path.read_bytes()
    assert b[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">II", b[16:24]) == (640, 360)
    n = struct.unpack(">I", b[33:37])[0]
    raw = zlib.decompress(b[41:41 + n])
    assert len(raw) == 360 * (1 + 640 * 3) and raw[1:1 + 640 * 3] == img[0].tobytes()
