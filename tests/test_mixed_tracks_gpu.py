"""BASELINE configs[4] -- track.json and big_track.json in ONE batch (car_env.py:621-628: every env may be reset onto its own
track; train.py:159,173-195) -- in the benchmarked dtype (F32: float32 selection, float64 refinement) against the float64 CPU
oracle, through every path a mixed batch can take:

  * the persistent rollout kernel's table-driven fast modes (a workgroup stages ONE track): 32768 envs (the per-rank shard of
    configs[4]: big form, 128 envs per workgroup), 65536 (256 per workgroup), 4096 (small form, 16 per workgroup);
  * its generic mode (a workgroup straddles the two tracks: every wave reads its own track's tables from global memory);
  * the per-step kernels K5 + K1 (bitwise the persistent kernel wherever both run), K1 with its per-wave waterfall over the
    distinct track ids when the tracks are interleaved env by env -- and, since round 6, pc_rollout on that layout too (the big
    form's generic mode with the same waterfall around its env step), F32 and F64 handles.

Bars as in test_rollout_baseline_gpu.py: observations within one float32 ulp of the oracle's, >= 99.99 % of the entries
bit-equal; rewards / flags exact on every env still on the oracle's trajectory; an env may leave it only at a step whose
threshold margin |d - 10 px| is <= 1e-9 px.  Every launch option is set through the trainer's own handles
(PPOConfig.rollout_form / policy_split -> pc_env_set_option / pc_policy_create): the library has no process-wide state."""
import dataclasses

import numpy as np
import pytest
import torch

from ppo_car_amd.ppo import PPOConfig, Trainer
from conftest import TRACKS
from test_rollout_baseline_gpu import _oracle_replay_check, _snap, strided_population

pytestmark = pytest.mark.gpu

MIXED = [TRACKS["track"], TRACKS["big_track"]]


def _track_ids(n_envs, interleave):
    i = np.arange(n_envs)
    return i % 2 if interleave else np.minimum((i // 32 * 32) * 2 // n_envs, 1)     # ppo.Trainer's two layouts


def _replay_per_track(cfg, snaps, first, n_envs, interleave, what, limit=1024):
    tid = _track_ids(n_envs, interleave)
    # one env out of every 32-env wave (rotating offsets; with interleaved tracks the offsets alternate between the tracks)
    sel = strided_population(n_envs, per_wave=(2 if interleave else 1) * (4 if n_envs < 4096 else 1), limit=limit)
    out = {}
    for k, path in enumerate(MIXED):
        mine = sel[tid[sel] == k]
        assert len(mine) >= 16, (what, k)
        out[k] = _oracle_replay_check(dataclasses.replace(cfg, track=path), snaps, first, f"{what} track {k}", sel=mine)
    return out


@pytest.mark.parametrize("n_envs,form,n_steps,want_kernel", [
    (32768, -1, 256, "fast, big form, 128 envs per workgroup (the per-rank shard of configs[4])"),
    (65536, -1, 128, "fast, big form, 256 envs per workgroup"),
    (4096, -1, 512, "fast, small form, 16 envs per workgroup"),
    (1000, 0, 256, "generic: the halves meet inside a workgroup"),
    (1000, 1, 256, "fast, small form, 32 envs per workgroup (forced)"),
], ids=["shard_32768", "65536", "4096_small_form", "1000_generic_big_form", "1000_small_form"])
def test_mixed_tracks_f32_halves_rollout_vs_step_kernels_and_oracle(n_envs, form, n_steps, want_kernel):
    res, first = {}, None
    for mode in ("mega", "steps"):
        cfg = PPOConfig(n_envs=n_envs, n_steps=n_steps, num_rays=16, track=MIXED, rollout_kernel=mode, use_graphs=False, seed=13,
                        env_dtype="f32", rollout_form=form, policy_split=-1 if form < 0 else form & 1)
        tr = Trainer(cfg, device="cuda")
        assert tr.envs.get_option("rollout_form") == form
        if first is None:
            first = tr.next_obs.clone()
        tr.rollout()
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager"), want_kernel
        res[mode] = _snap(tr)
        res[mode + "_state"] = tr.envs.get_state()
        tr.close()
        del tr
    for i, (a, b) in enumerate(zip(res["mega"], res["steps"])):
        assert torch.equal(a, b), f"buffer {i} differs between pc_rollout and the per-step kernels ({want_kernel})"
    for k in res["mega_state"]:
        assert np.array_equal(res["mega_state"][k], res["steps_state"][k]), k
    half = n_envs // 2 // 32 * 32
    assert res["mega_state"]["next_gate"][:half].max() < 45            # track.json has 45 gates, big_track.json 55
    out = _replay_per_track(cfg, res["mega"], first, n_envs, False, f"mixed N={n_envs} form={form}")
    for k, (worst, ties, alive) in out.items():
        print(f"mixed halves N={n_envs} ({want_kernel}) track {k}: obs max err {worst:.2e}, near-tie flips {ties}, on trajectory {alive:.3f}")
    del res
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype,fast", [("f32", 1), ("f32", 3), ("f32", 2), ("f64", 1), ("f64", 3), ("f64", 2)],
                         ids=["f32_deinterleaved_by_wave", "f32_two_track_passes", "f32_generic_mode", "f64_deinterleaved_literal", "f64_two_track_passes_literal", "f64_K9d"])
@pytest.mark.parametrize("n_envs", [2048, 20000, 32768, 65536])
def test_mixed_tracks_interleaved_rollout_vs_step_kernels_and_oracle(n_envs, dtype, fast):
    """track_id = i & 1 (SURVEY 8(d) C4's second variant; car_env.py:621-628 lets every env sit on its own track): every wave holds both
    tracks.  pc_rollout runs it in the BIG form with the env step once per track present in a wave (K1's waterfall, which the per-step
    path takes too): F32 handles in the two-track FAST form (rollout_kernel<6, 9, 2, 6>: both tracks' tables in LDS, the table-driven
    step per pass; F64 handles: its literal form) or, with the fast modes' track layouts switched off, in the generic mode / K9d.  Every buffer bitwise
    the per-step kernels' (unsplit policy arithmetic: the big form's), then the oracle per track -- one float32 ulp / events exact for
    F32 handles, every bit for F64 handles."""
    res, first = {}, None
    for mode in ("mega", "steps"):
        cfg = PPOConfig(n_envs=n_envs, n_steps=192, num_rays=16, track=MIXED, track_interleave=True, rollout_kernel=mode,
                        use_graphs=False, seed=17, env_dtype=dtype, policy_split=0, rollout_fast=fast)
        tr = Trainer(cfg, device="cuda")
        if first is None:
            first = tr.next_obs.clone()
        for _ in range(2):
            tr.rollout()
            tr.buffer.ptr = 0
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager")
        if mode == "mega":
            # fast 1: `i & 1` splits every block of 64 / 32 envs evenly -- the block's two waves de-interleave it (rollout_kernel's mode 7: one pass per
            # wave); 3: the per-track passes inside every wave (mode 6: what an unevenly interleaved batch takes); 2: the generic mode
            m = "m" if (fast in (1, 3) and 8192 < n_envs <= 32768) else ""      # 16 envs per wave between 8193 and 32768 envs, as for every other layout
            assert tr.envs.last_rollout_kernel() == (f"K9{m}" if dtype == "f32" else (f"K9{m}-literal" if fast in (1, 3) else "K9d-selector"))
        res[mode] = _snap(tr)
        res[mode + "_state"] = tr.envs.get_state()
        tr.close()
        del tr
    for i, (a, b) in enumerate(zip(res["mega"], res["steps"])):
        assert torch.equal(a, b), f"buffer {i} differs between pc_rollout and the per-step kernels (interleaved tracks, {dtype})"
    for k in res["mega_state"]:
        assert np.array_equal(res["mega_state"][k], res["steps_state"][k]), k
    assert res["mega_state"]["next_gate"][0::2].max() < 45            # track.json has 45 gates, big_track.json 55
    # one rollout from reset against the oracle, per track
    cfg = dataclasses.replace(cfg, rollout_kernel="mega", n_steps=256)
    tr = Trainer(cfg, device="cuda")
    first = tr.next_obs.clone()
    tr.rollout()
    torch.cuda.synchronize()
    assert tr.rollout_mode == "mega"
    snaps = _snap(tr)
    tr.close()
    tid = _track_ids(n_envs, True)
    sel = strided_population(n_envs, per_wave=2 * (4 if n_envs < 4096 else 1), limit=1024)
    for k, path in enumerate(MIXED):
        mine = sel[tid[sel] == k]
        assert len(mine) >= 16
        worst, ties, alive = _oracle_replay_check(dataclasses.replace(cfg, track=path), snaps, first, f"interleaved N={n_envs} {dtype} track {k}", sel=mine,
                                                  exact=dtype == "f64")
        print(f"mixed interleaved N={n_envs} {dtype} track {k}: obs max err {worst:.2e}, near-tie flips {ties}, on trajectory {alive:.3f}")
    del res
    torch.cuda.empty_cache()
