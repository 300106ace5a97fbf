"""Summarise rocprofv3 --pmc passes of the env-step kernel into profiles/ (developer tool).
usage: python tools/pmc_summary.py <pmc_dir> <tag> <n_envs> <num_rays> <dtype> [kernel_substring] [n_steps]
kernel_substring defaults to env_step_kernel; with rollout_kernel pass n_steps (bytes are per launch = per rollout)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc_dir, tag, N, n, dtype = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
KSUB = sys.argv[6] if len(sys.argv) > 6 else "env_step_kernel"
T = int(sys.argv[7]) if len(sys.argv) > 7 else 0
agg = collections.defaultdict(list)
dur = []
for f in sorted(glob.glob(os.path.join(pmc_dir, "*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if KSUB in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(os.path.join(pmc_dir, "*", "*", "*kernel_trace.csv"))):
    for r in csv.DictReader(open(f)):
        if KSUB in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
mean = {k: sum(v) / len(v) for k, v in agg.items()}
fetch_kb, write_kb = mean.get("FETCH_SIZE"), mean.get("WRITE_SIZE")
# MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads; WRITE_SIZE is exact.
# Calibration on this kernel's known byte count: reads are 56 B/env (32 B state + 16 B counters + 8 B action).
hbm = (2 * fetch_kb + write_kb) * 1024 if fetch_kb is not None and write_kb is not None else None
out = {"kernel": KSUB, "n_steps_per_launch": T or 1, "n_envs": N, "num_rays": n, "dtype": dtype, "launches": len(next(iter(agg.values()))) if agg else 0,
       "counters_mean_per_launch": mean, "duration_us_under_pmc_mean": sum(dur) / len(dur) if dur else None,
       "hbm_bytes_per_launch": hbm, "fetch_kb_raw": fetch_kb, "write_kb": write_kb,
       "known_bytes_per_env": ({"read": 56, "write": 4 * (6 + {12: 12, 16: 17, 32: 33}.get(n, n)) + 48 + 12} if not T else
                               {"read_per_rollout": 48 + 4 * (6 + {12: 12, 16: 17, 32: 33}.get(n, n)),
                                "write_per_env_step": 4 * (6 + {12: 12, 16: 17, 32: 33}.get(n, n)) + 24}),
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at 1/2); WRITE_SIZE as is. "
               "SQ_* counters are summed over all waves; SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles."}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}.json"), "w"), indent=1)
tf_path = os.path.join(ROOT, "profiles", "k1_traffic.json")
tf = json.load(open(tf_path)) if os.path.exists(tf_path) else {}
if hbm is not None:
    tf[(f"rollout_{dtype}_n{n}_N{N}_T{T}" if T else f"{dtype}_n{n}_N{N}")] = {"hbm_bytes_per_launch": hbm, "source": f"profiles/{tag}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
    json.dump(tf, open(tf_path, "w"), indent=1)
print(json.dumps(out, indent=1))
