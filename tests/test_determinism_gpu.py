"""Launch-to-launch determinism of the persistent rollout kernels at the BENCHMARKED shapes: the same rollout (same env state, same Philox
counters, same weights) launched 200 times; every buffer of every launch must equal the first launch's bit for bit.

Why this is a test and not only a developer probe: round 5 built a 33-ray variant (rollout_kernel<10, 17, 2, 4>, the chain-packed sweep in
three passes) that differed from the per-step kernels in ~1 observation entry of 1e9 -- nondeterministically, only with two waves per
SIMD, only in the sweep passes that consumed values kept in VGPRs from the first pass through packed-fp32 instructions
(profiles/r6_cfg2_packed_rootcause.md).  The kernels have no inter-wave communication after staging, so ANY launch-to-launch difference is
a hazard of that class.  Four launches (what tools/determinism_check.py ran) cannot see a 1e-9 event; 200 launches x 2e9 entries can."""
import pytest
import torch

from ppo_car_amd.ppo import PPOConfig, Trainer
from conftest import TRACKS

pytestmark = pytest.mark.gpu

MIXED = [TRACKS["track"], TRACKS["big_track"]]
SHAPES = {
    "target": (dict(n_envs=65536, n_steps=1024, num_rays=16, track=TRACKS["big_track"]), "K9", 200),
    "cfg1": (dict(n_envs=4096, n_steps=1024, num_rays=16, track=TRACKS["big_track"]), "K9s", 200),
    "cfg2": (dict(n_envs=65536, n_steps=128, num_rays=32, track=TRACKS["big_track"]), "K9", 400),
    "cfg4": (dict(n_envs=32768, n_steps=1024, num_rays=16, track=MIXED), "K9m", 200),
    "cfg4i": (dict(n_envs=32768, n_steps=512, num_rays=16, track=MIXED, track_interleave=True), "K9m", 200),      # the two-track fast form
    "cfg4i_65536": (dict(n_envs=65536, n_steps=256, num_rays=16, track=MIXED, track_interleave=True), "K9", 200),
    "cfg4i_f64": (dict(n_envs=32768, n_steps=256, num_rays=16, track=MIXED, track_interleave=True, env_dtype="f64"), "K9m-literal", 200),
    "target_f64": (dict(n_envs=65536, n_steps=256, num_rays=16, track=TRACKS["big_track"], env_dtype="f64"), "K9-literal", 200),
    "cfg2_f64": (dict(n_envs=65536, n_steps=128, num_rays=32, track=TRACKS["big_track"], env_dtype="f64"), "K9-literal", 200),
}


@pytest.mark.parametrize("name", list(SHAPES))
def test_persistent_rollout_is_bit_identical_launch_after_launch(name):
    kw, kernel, reps = SHAPES[name]
    tr = Trainer(PPOConfig(seed=7, rollout_kernel="mega", **kw), device="cuda")
    for _ in range(2):                      # a policy that has moved and envs spread over the track
        tr.run_epoch(sync=False)
    torch.cuda.synchronize()
    assert tr.rollout_mode == "mega" and tr.envs.last_rollout_kernel() == kernel
    st = tr.envs.get_state()
    keep = [t.clone() for t in (tr.next_obs, tr.next_term, tr.next_trunc, tr.rng_base)]
    b = tr.buffer
    outs = (b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.logprob_buf, b.term_buf, b.trunc_buf, tr.next_obs, tr.next_term, tr.next_trunc)
    first, entries = None, 0
    for r in range(reps):
        tr.envs.set_state(**st)
        for dst, src in zip((tr.next_obs, tr.next_term, tr.next_trunc, tr.rng_base), keep):
            dst.copy_(src)
        tr.rollout()
        tr.buffer.ptr = 0
        if first is None:
            first = [x.clone() for x in outs]
        else:
            for i, (a, c) in enumerate(zip(first, outs)):
                assert torch.equal(a, c), f"{name}: launch {r} differs from launch 0 in buffer {i} ({int((a != c).sum())} entries)"
        entries += sum(x.numel() for x in outs)
    torch.cuda.synchronize()
    print(f"{name}: {reps} launches of {kernel}, {entries:.2e} entries, all bit-identical")
    tr.close()
    del first
    torch.cuda.empty_cache()
