"""Developer tool (GPU box): soak a persistent rollout kernel against the per-step kernels, launch after launch, until a difference shows,
and then find out WHICH side was wrong and WHERE the wrong value lives.

Written for round 5's unexplained event (profiles/r5_ab_experiments.txt): the chain-packed 33-ray variant rollout_kernel<10, 17, 2, 4>
gave ONE observation entry of 3.3e8 different from the per-step kernels in one run of two.  The record of the hunt -- and of the shipped
kernels' soaks -- is profiles/r6_cfg2_packed_rootcause.md (the -DPC_EXP_* developer flags it names exist in commit 171d065 only).

    python tools/soak_rollout.py <package dir with the build under test> [--launches N] [--n-envs 65536] [--n-steps 128] [--rays 32]
                                     [--fast 1] [--dtype f32] [--out FILE]

Two trainers of one seed advance in lockstep: A = pc_rollout (the build under test), B = the per-step kernels K5 + K1 of the same library.
After every launch all ten outputs are compared on the device.  On a difference the tool
  * records every differing entry (buffer, step, env, column, both values) and whether anything DOWNSTREAM of it differs (value / log-prob /
    action of the next step, which the policy computes from the LDS copy of the observation -- a wrong value only in the global buffer
    leaves them untouched);
  * restores both trainers to the state before the launch and repeats the launch twice on each side: the side that does not reproduce its
    own result is the nondeterministic one;
  * continues from B's state.
One JSON line per event and a summary line at the end."""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("pkg")
ap.add_argument("--launches", type=int, default=500)
ap.add_argument("--n-envs", type=int, default=65536)
ap.add_argument("--n-steps", type=int, default=128)
ap.add_argument("--rays", type=int, default=32)
ap.add_argument("--fast", type=int, default=1)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--mixed", action="store_true", help="track.json + big_track.json in halves (configs[4]'s layout)")
ap.add_argument("--interleave", action="store_true", help="with --mixed: track_id = i & 1 instead of halves")
ap.add_argument("--form", type=int, default=-1, help="PPOConfig.rollout_form (2: the big form without the 1/den table in LDS)")
ap.add_argument("--train-every", type=int, default=25, help="one PPO update every k launches (both sides: the policy moves, identically)")
ap.add_argument("--repeats", type=int, default=10, help="on an event: repeat the rollout launch this many times from the saved state")
ap.add_argument("--max-events", type=int, default=5)
ap.add_argument("--out", default=None)
args = ap.parse_args()
sys.path.insert(0, os.path.abspath(args.pkg))
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ppo_car_amd.ppo import PPOConfig, Trainer  # noqa: E402

NAMES = ["obs_buf", "act_buf", "rew_buf", "val_buf", "logprob_buf", "term_buf", "trunc_buf", "next_obs", "next_term", "next_trunc"]


def bufs(tr):
    b = tr.buffer
    return [b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.logprob_buf, b.term_buf, b.trunc_buf, tr.next_obs, tr.next_term, tr.next_trunc]


def make(mode):
    cfg = PPOConfig(n_envs=args.n_envs, n_steps=args.n_steps, num_rays=args.rays, track=([f"{ROOT}/tracks/track.json", f"{ROOT}/tracks/big_track.json"] if args.mixed else f"{ROOT}/tracks/big_track.json"), rollout_kernel=mode,
                    use_graphs=False, seed=11, env_dtype=args.dtype, rollout_fast=args.fast, rollout_form=args.form, track_interleave=bool(args.interleave), bootstrap_value="fp32", policy_split=1 if args.n_envs <= 8192 else 0)
    return Trainer(cfg, device="cuda")


def save(tr):
    return dict(env=tr.envs.get_state(), next_obs=tr.next_obs.clone(), next_term=tr.next_term.clone(), next_trunc=tr.next_trunc.clone(),
                rng=tr.rng_base.clone())


def restore(tr, sd):
    tr.envs.set_state(**sd["env"])
    tr.next_obs.copy_(sd["next_obs"]); tr.next_term.copy_(sd["next_term"]); tr.next_trunc.copy_(sd["next_trunc"]); tr.rng_base.copy_(sd["rng"])


def launch(tr):
    tr.rollout()
    tr.buffer.ptr = 0
    torch.cuda.synchronize()
    return [t.clone() for t in bufs(tr)]


A, B = make("mega"), make("steps")
out = open(args.out, "a") if args.out else None
events, entries, t0, cols = [], 0, time.time(), {}
for it in range(args.launches):
    sd = save(B)
    ra, rb = launch(A), launch(B)
    assert A.rollout_mode == "mega", A.rollout_mode
    entries += sum(t.numel() for t in ra)
    diff = [i for i, (x, y) in enumerate(zip(ra, rb)) if not torch.equal(x, y)]
    if diff:
        ev = {"launch": it, "kernel": A.envs.last_rollout_kernel(), "buffers": [NAMES[i] for i in diff], "entries": []}
        for i in diff:
            idx = (ra[i] != rb[i]).nonzero()[:16].cpu().numpy()
            for row in idx:
                ev["entries"].append({"buffer": NAMES[i], "index": [int(v) for v in row], "rollout": float(ra[i][tuple(row)]), "per_step": float(rb[i][tuple(row)])})
        # who is nondeterministic?  the same launch again from the saved state: twice on the per-step side, `--repeats` times on the rollout side,
        # tallying WHICH entries of the observation buffer differ from the per-step kernels' each time
        rep = {}
        same = []
        for _ in range(2):
            restore(B, sd)
            again = launch(B)
            same.append(all(torch.equal(x, y) for x, y in zip(rb, again)))
        rep["per_step_reproduces_itself"] = same
        tally, n_bad, first_cols = {}, [], []
        for _ in range(args.repeats):
            restore(A, sd)
            again = launch(A)
            bad = (again[0] != rb[0]).nonzero()
            n_bad.append(int(bad.shape[0]))
            # only FIRST differences per env count as primary (later ones follow from a diverged trajectory)
            if bad.shape[0]:
                bb = bad.cpu().numpy()
                seen = {}
                for t_, e_, c_ in bb:
                    if e_ not in seen or t_ < seen[e_][0]:
                        seen[e_] = (int(t_), int(c_))
                for e_, (t_, c_) in seen.items():
                    tally[(t_, int(e_), c_)] = tally.get((t_, int(e_), c_), 0) + 1
                    cols[c_] = cols.get(c_, 0) + 1
        rep["rollout_repeats"] = args.repeats
        rep["differing_obs_entries_per_repeat"] = n_bad
        rep["primary_entries_seen_k_times"] = sorted(((list(k), v) for k, v in tally.items()), key=lambda kv: -kv[1])[:24]
        ev.update(rep)
        ev["entries"] = ev["entries"][:12]
        events.append(ev)
        print(json.dumps(ev), flush=True)
        if out:
            out.write(json.dumps(ev) + "\n"); out.flush()
        # continue from identical states and buffers (B's)
        restore(B, sd)
        rb = launch(B)
        restore(A, save(B))
        for x, y in zip(bufs(A), bufs(B)):
            x.copy_(y)
        if len(events) >= args.max_events:
            break
    if args.train_every and (it + 1) % args.train_every == 0:
        for tr in (A, B):
            tr.buffer.ptr = tr.cfg.n_steps
            tr.update()
        torch.cuda.synchronize()
        assert torch.equal(A.learner.flat_param, B.learner.flat_param), 'the two trainers left lockstep'
    if it % 50 == 49:
        print(f"# {it + 1} launches, {entries:.3e} entries compared, {len(events)} events, {time.time() - t0:.0f} s", flush=True)
summary = {"summary": True, "pkg": args.pkg, "kernel": A.envs.last_rollout_kernel(), "launches": args.launches, "n_envs": args.n_envs, "n_steps": args.n_steps,
           "rays": args.rays, "mixed": args.mixed, "interleave": bool(args.interleave), "fast": args.fast, "dtype": args.dtype, "entries_compared": entries, "events": len(events), "obs_columns_of_primary_differences": {str(k): v for k, v in sorted(cols.items())}, "seconds": time.time() - t0}
print(json.dumps(summary), flush=True)
if out:
    out.write(json.dumps(summary) + "\n")
A.close(); B.close()
