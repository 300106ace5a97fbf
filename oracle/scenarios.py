"""oracle.scenarios -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/__init__.py).

Starting states that drive CarEnv.step into the branches a freshly initialised policy never reaches -- the lap wrap (car_env.py:730-737),
the time limit and its `elif` (car_env.py:746-750), hundreds of turns of heading drift (car_env.py:440-442) -- for the parity tests of the
persistent rollout kernels (tests/test_rollout_rare_branches_gpu.py) and bench.py's second parity leg; and the loader of the trained
policy fixture (tests/golden/policy_trained.npz, written by tools/make_policy_fixture.py)."""
import os

import numpy as np

from . import Track, ray_distance

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_trained_policy(agent):
    """tests/golden/policy_trained.npz (150 epochs on big_track at 16 -> 17 rays, ~3 laps per episode) into an Agent(23, 9)"""
    import torch
    f = np.load(os.path.join(_ROOT, "tests", "golden", "policy_trained.npz"))
    sd = {k: torch.from_numpy(f[k.replace(".", "_")]) for k in agent.state_dict()}
    with torch.no_grad():
        agent.load_state_dict(sd)


def rot_after(start_rot, k):
    """the reference's heading after |k| turns in one direction: a running float64 sum (car_env.py:440-442), not start + 5 k"""
    r = np.float64(start_rot)
    for _ in range(abs(int(k))):
        r = r + (5.0 if k > 0 else -5.0)
    return float(r)


def injected_state(track_path, idx):
    """State arrays for the envs `idx` (their batch indices: the pattern depends on them) on one track.  Five groups by i % 5:
      0  on the approach to the LAST gate (15 px before it, 8 px per step towards it), next_gate = G - 1: the lap wrap
      1  the same with time_step 997 .. 999: lap and time limit in one episode's last steps
      2  start pose, heading start_rot +- 80 .. 88 turns, time_step 997 .. 999: the heading wrap past +-72 turns; truncation
      3  start pose, heading +- 900 .. 990 turns with time_step 995 .. 999: the deep rows of the rotation table
      4  12 .. 40 px from the wall the start pose looks at, 9 px per step towards it, time_step 997 .. 999: a crash in the step in which the
         time limit falls due (terminated wins, Q7) for a part of them"""
    t = Track(track_path)
    n = len(idx)
    i = np.asarray(idx, np.int64)
    grp = i % 5
    j = i // 5
    px = np.full(n, t.start_x); py = np.full(n, t.start_y)
    vx = np.zeros(n); vy = np.zeros(n)
    rot = np.full(n, t.start_rot)
    time_step = np.zeros(n, np.int64); next_gate = np.zeros(n, np.int64); passed = np.zeros(n, np.int64)
    # groups 0 / 1: the approach to the last gate, along the line from the gate before it
    g1 = t.gates[t.G - 1]; g0 = t.gates[t.G - 2]
    m1 = np.array([(g1[0] + g1[2]) / 2, (g1[1] + g1[3]) / 2]); m0 = np.array([(g0[0] + g0[2]) / 2, (g0[1] + g0[3]) / 2])
    u = (m1 - m0) / np.linalg.norm(m1 - m0)
    k_dir = int(np.rint((np.degrees(np.arctan2(u[1], u[0])) - t.start_rot) / 5.0))
    a = (grp == 0) | (grp == 1)
    back = 15.0 + (j % 7)                                     # 15 .. 21 px before the gate
    px[a] = (m1[0] - back * u[0])[a]
    py[a] = (m1[1] - back * u[1])[a]
    vx[a] = 8.0 * u[0]; vy[a] = 8.0 * u[1]
    rot[a] = rot_after(t.start_rot, k_dir)
    next_gate[a] = t.G - 1
    passed[a] = t.G - 1 + t.G * (j[a] % 3)
    time_step[grp == 0] = 300 + (j[grp == 0] % 50)
    time_step[grp == 1] = 997 + (j[grp == 1] % 3)
    # groups 2 / 3: heading drift
    for g, ks in ((2, [80, -80, 84, -88]), (3, [900, -900, 990, -990, 950, -975])):
        rots = [rot_after(t.start_rot, k) for k in ks]
        m = grp == g
        rot[m] = np.asarray(rots)[j[m] % len(ks)]
    time_step[grp == 2] = 997 + (j[grp == 2] % 3)
    time_step[grp == 3] = 995 + (j[grp == 3] % 5)
    # group 4: towards the wall ahead of the start pose
    d0 = ray_distance(t.start_x, t.start_y, t.start_rot, t.walls)
    c, s = np.cos(np.radians(t.start_rot)), np.sin(np.radians(t.start_rot))
    m = grp == 4
    d = 12.0 + (j[m] % 29)
    px[m] = t.start_x + (d0 - d) * c; py[m] = t.start_y + (d0 - d) * s
    vx[m] = 9.0 * c; vy[m] = 9.0 * s
    time_step[m] = 997 + ((j[m] // 29) % 3)
    return dict(px=px, py=py, vx=vx, vy=vy, rot=rot, time_step=time_step, next_gate=next_gate, passed=passed)
