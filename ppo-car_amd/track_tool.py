"""Track JSON authoring / validation, headless (the counterpart of the reference's interactive
track_editor.py).  The schema is the one track_editor.py writes and CarEnv.load_track reads (car_env.py:535-567):

    outer_track_points, inner_track_points : [[x, y], ...]  normalised to the 1280 x 720 window, 4 decimals
                                             (track_editor.py:62-75), CLOSED: the editor appends the first point again
                                             when the loop is finished (next_mode, track_editor.py:216-221)
    reward_gates                           : flat list of points, consecutive pairs are one gate; the first gate is the
                                             start / finish line
    initial_position : [x, y]              initial_angle : float

    python -m ppo_car_amd.track_tool validate  tracks/big_track.json
    python -m ppo_car_amd.track_tool info      tracks/big_track.json
    python -m ppo_car_amd.track_tool normalise in.json out.json
    python -m ppo_car_amd.track_tool make-oval out.json --points 64 --gates 40

No GPU needed (the loader behind `info` is the C-ABI's host-side parser, the one the env itself uses).
"""
import argparse
import json
import math
import sys

import numpy as np

KEYS = ("outer_track_points", "inner_track_points", "reward_gates", "initial_position", "initial_angle")
WIDTH, HEIGHT = 1280.0, 720.0


# ---- schema ---------------------------------------------------------------------------------------------
def _is_point(p):
    return isinstance(p, (list, tuple)) and len(p) == 2 and all(isinstance(v, (int, float)) and math.isfinite(v) for v in p)


def check_schema(d):
    """Problems as strings; an empty list means CarEnv.load_track (and pc_track_load_json) will accept the file."""
    out = []
    if not isinstance(d, dict):
        return ["top level is not an object"]
    for k in KEYS:
        if k not in d:
            out.append(f"missing key {k!r}")
    if out:
        return out
    for k in ("outer_track_points", "inner_track_points", "reward_gates"):
        v = d[k]
        if not isinstance(v, list) or not all(_is_point(p) for p in v):
            out.append(f"{k}: not a list of [x, y] numbers")
        elif any(not (0.0 <= p[0] <= 1.0 and 0.0 <= p[1] <= 1.0) for p in v):
            out.append(f"{k}: coordinates outside the normalised window [0, 1]")
    if out:
        return out
    for k in ("outer_track_points", "inner_track_points"):
        v = d[k]
        if len(v) < 4:
            out.append(f"{k}: a closed loop needs at least 3 distinct points plus the repeated first one")
        elif list(v[0]) != list(v[-1]):
            out.append(f"{k}: loop not closed (the editor repeats the first point at the end)")
        if any(list(a) == list(b) for a, b in zip(v, v[1:])):
            out.append(f"{k}: zero-length segment (two equal consecutive points)")
    g = d["reward_gates"]
    if len(g) < 2:
        out.append("reward_gates: no gate")
    if len(g) % 2:
        out.append("reward_gates: odd number of points (gates are consecutive pairs)")
    if not _is_point(d["initial_position"]):
        out.append("initial_position: not [x, y]")
    if not isinstance(d["initial_angle"], (int, float)) or not math.isfinite(d["initial_angle"]):
        out.append("initial_angle: not a finite number")
    for k in ("outer_track_points", "inner_track_points", "reward_gates"):
        if any(round(c, 4) != c for p in d[k] for c in p):
            out.append(f"{k}: coordinates with more than 4 decimals (the editor rounds to 4; `normalise` fixes this)")
    return out


# ---- geometry -------------------------------------------------------------------------------------------
def _seg_intersect(p1, p2, p3, p4, touching=False):
    """Proper intersection of the open segments p1p2 and p3p4 (touching=True: an endpoint on the other segment counts)."""
    def orient(a, b, c):
        return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
    d1, d2, d3, d4 = orient(p3, p4, p1), orient(p3, p4, p2), orient(p1, p2, p3), orient(p1, p2, p4)
    if touching:
        return d1 * d2 <= 0 and d3 * d4 <= 0 and (d1 != 0 or d2 != 0 or d3 != 0 or d4 != 0)
    return d1 * d2 < 0 and d3 * d4 < 0


def _inside(poly, p):
    """Even-odd rule; poly closed (first point repeated)."""
    x, y, c = p[0], p[1], False
    for (x1, y1), (x2, y2) in zip(poly, poly[1:]):
        if (y1 > y) != (y2 > y) and x < x1 + (y - y1) * (x2 - x1) / (y2 - y1):
            c = not c
    return c


def check_geometry(d):
    """Problems a schema-valid file can still have: self-intersecting walls, inner loop not inside the outer one, start or
    gates outside the corridor, a gate that does not span the corridor."""
    out = []
    outer = [(p[0] * WIDTH, p[1] * HEIGHT) for p in d["outer_track_points"]]
    inner = [(p[0] * WIDTH, p[1] * HEIGHT) for p in d["inner_track_points"]]
    for name, loop in (("outer_track_points", outer), ("inner_track_points", inner)):
        segs = list(zip(loop, loop[1:]))
        for i in range(len(segs)):
            for j in range(i + 2, len(segs)):
                if i == 0 and j == len(segs) - 1:
                    continue        # neighbours through the closing point
                if _seg_intersect(*segs[i], *segs[j]):
                    out.append(f"{name}: segments {i} and {j} cross")
    for a in zip(outer, outer[1:]):
        for b in zip(inner, inner[1:]):
            if _seg_intersect(*a, *b):
                out.append("outer and inner walls cross")
                break
    if not all(_inside(outer, p) for p in inner[:-1]):
        out.append("inner loop is not inside the outer loop")

    def in_corridor(p):
        return _inside(outer, p) and not _inside(inner, p)
    start = (d["initial_position"][0] * WIDTH, d["initial_position"][1] * HEIGHT)
    if not in_corridor(start):
        out.append("initial_position is not between the walls")
    g = [(p[0] * WIDTH, p[1] * HEIGHT) for p in d["reward_gates"]]
    walls = list(zip(outer, outer[1:])), list(zip(inner, inner[1:]))
    for k in range(0, len(g) - 1, 2):
        a, b = g[k], g[k + 1]
        mid = ((a[0] + b[0]) / 2, (a[1] + b[1]) / 2)
        if not in_corridor(mid):
            out.append(f"gate {k // 2}: midpoint outside the corridor")
            continue
        # a gate drawn from wall to wall: each end within a few pixels of (or beyond) one of the two loops
        def reaches(seglist):
            return any(_seg_intersect(a, b, *s, touching=True) for s in seglist) or min(_pt_seg_dist(e, s) for e in (a, b) for s in seglist) < 12.0
        if not (reaches(walls[0]) and reaches(walls[1])):
            out.append(f"gate {k // 2}: does not span the corridor (a car can pass beside it)")
    return out


def _pt_seg_dist(p, s):
    (x1, y1), (x2, y2) = s
    dx, dy = x2 - x1, y2 - y1
    t = max(0.0, min(1.0, ((p[0] - x1) * dx + (p[1] - y1) * dy) / (dx * dx + dy * dy)))
    return math.hypot(p[0] - x1 - t * dx, p[1] - y1 - t * dy)


# ---- authoring ------------------------------------------------------------------------------------------
def normalise(d):
    """What track_editor.py does on save: 4-decimal coordinates, loops closed by repeating the first point; also drops
    consecutive duplicates (zero-length walls)."""
    out = dict(d)
    for k in ("outer_track_points", "inner_track_points", "reward_gates"):
        pts = [[round(float(p[0]), 4), round(float(p[1]), 4)] for p in d[k]]
        if k != "reward_gates":
            pts = [p for i, p in enumerate(pts) if i == 0 or p != pts[i - 1]]
            if pts and pts[0] != pts[-1]:
                pts.append(list(pts[0]))
        out[k] = pts
    out["initial_position"] = [round(float(v), 4) for v in d["initial_position"]]
    out["initial_angle"] = float(d["initial_angle"])
    return out


def make_oval(n_points=32, n_gates=24, rx=0.42, ry=0.40, width=0.13, wobble=0.0, seed=0):
    """A synthetic closed circuit with n_points wall points per loop -- the knob of the segment-count stress configs
    (BASELINE configs[4]): walls = 2 * n_points, vertex-chain length = 2 * (n_points + 1)."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0.0, 2.0 * np.pi, n_points, endpoint=False)
    r = 1.0 + wobble * rng.uniform(-1.0, 1.0, n_points)

    def loop(sx, sy):
        pts = [[round(0.5 + sx * r[i] * math.cos(th[i]), 4), round(0.5 + sy * r[i] * math.sin(th[i]), 4)] for i in range(n_points)]
        return pts + [list(pts[0])]
    outer, inner = loop(rx, ry), loop(rx - width, ry - width * WIDTH / HEIGHT * 0.6)
    gates = []
    for j in range(n_gates):   # gate 0 = start / finish line; gates ordered in the driving direction (increasing angle)
        a = 2.0 * np.pi * j / n_gates + 0.37 * 2.0 * np.pi / n_points   # (off the wall vertices)
        for s in (1.04, 0.90):  # from just outside the outer loop to just inside the inner one
            sx = (rx if s > 1 else rx - width) * s
            sy = (ry if s > 1 else ry - width * WIDTH / HEIGHT * 0.6) * s
            gates.append([round(0.5 + sx * math.cos(a), 4), round(0.5 + sy * math.sin(a), 4)])
    a0 = -2.0 * np.pi * 0.5 / n_gates + 0.37 * 2.0 * np.pi / n_points  # half a gate spacing BEFORE gate 0 (the first gate to
    #                                                                    pass: next_gate_index starts at 0), heading along the tangent
    mx, my = rx - width / 2, ry - width * WIDTH / HEIGHT * 0.3
    start = [round(0.5 + mx * math.cos(a0), 4), round(0.5 + my * math.sin(a0), 4)]
    ang = math.degrees(math.atan2(my * HEIGHT * math.cos(a0), -mx * WIDTH * math.sin(a0)))   # the heading is in degrees (car_env.py:426)
    return {"outer_track_points": outer, "inner_track_points": inner, "reward_gates": gates, "initial_position": start,
            "initial_angle": ang}


def summary(d, path=None):
    """Counts as the env sees them; with a path also what the C-ABI loader reports (must agree)."""
    n_out, n_in = len(d["outer_track_points"]) - 1, len(d["inner_track_points"]) - 1
    walls, gates = n_out + n_in, len(d["reward_gates"]) // 2
    n_vtx = walls + 2                              # two chains, each with one extra opening vertex
    n_vtx_padded = (n_vtx + 3) // 4 * 4            # the sweep walks vertex groups of four
    info = {"walls": walls, "gates": gates, "chain_vertices": n_vtx, "chain_vertices_padded": n_vtx_padded,
            "inv_den_table_bytes": 361 * n_vtx_padded * 4,
            # the persistent rollout kernel keeps the 1/den table in LDS when it fits next to the 62.5 KB policy image,
            # the observation tile and the small tables; else the sweep computes 1/den itself
            "inv_den_table_fits_lds_at_16_rays": 361 * n_vtx_padded * 4 <= 160 * 1024 - 115 * 1024}
    if path is not None:
        from .env import Track
        t = Track(path)
        info["loader"] = {"walls": t.n_walls, "gates": t.n_gates, "start": [t.start_x, t.start_y, t.start_angle]}
        if (t.n_walls, t.n_gates) != (walls, gates):
            raise AssertionError(f"loader disagrees with the file: {info['loader']} vs {walls} walls, {gates} gates")
    return info


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m ppo_car_amd.track_tool", description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("validate", "info"):
        p = sub.add_parser(name)
        p.add_argument("track")
    p = sub.add_parser("normalise")
    p.add_argument("src")
    p.add_argument("dst")
    p = sub.add_parser("make-oval")
    p.add_argument("dst")
    p.add_argument("--points", type=int, default=32)
    p.add_argument("--gates", type=int, default=24)
    p.add_argument("--wobble", type=float, default=0.0)
    p.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    if a.cmd == "make-oval":
        d = make_oval(a.points, a.gates, wobble=a.wobble, seed=a.seed)
        json.dump(d, open(a.dst, "w"))
        probs = check_schema(d) + check_geometry(d)
        print(json.dumps({"written": a.dst, "problems": probs, **summary(d)}))
        return 1 if probs else 0
    if a.cmd == "normalise":
        d = normalise(json.load(open(a.src)))
        json.dump(d, open(a.dst, "w"))
        print(json.dumps({"written": a.dst, "problems": check_schema(d)}))
        return 0
    d = json.load(open(a.track))
    probs = check_schema(d)
    if not probs:
        probs += check_geometry(d)
    if a.cmd == "validate":
        print(json.dumps({"track": a.track, "ok": not probs, "problems": probs}))
        return 1 if probs else 0
    print(json.dumps({"track": a.track, "problems": probs, **(summary(d, a.track) if not check_schema(d) else {})}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
