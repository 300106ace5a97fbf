#!/usr/bin/env python3
"""Static instruction breakdown of one kernel from hipcc's -save-temps assembly.

    hipcc --offload-arch=gfx950 ... -save-temps -c ppocar.hip      # writes *-gfx950.s
    python tools/isa_breakdown.py ppocar-hip-amdgcn-amd-amdhsa-gfx950.s 'rollout_kernelILi6ELi9ELi2E' [--blocks]

Prints, per basic block (label to label) of the kernel, the count of VALU / packed VALU / transcendental / f64 /
MFMA / SALU / SMEM / LDS / VMEM / lane-spill (v_readlane, v_writelane) / waitcnt instructions, marks loop heads
(targets of backward branches) and shows `; PCMARK <name>` comments (emitted by the kernel's PC_MARK() macro in a
-DPPOCAR_MARKERS build) so that blocks can be attributed to phases.  With trip counts supplied as
--trip LABEL=N (e.g. the wall-vertex loop: 7 groups for big_track) it sums a weighted per-step total.
"""
import argparse
import collections
import re
import sys

CLASSES = ["valu", "pk", "trans", "f64", "mfma", "salu", "smem", "lds", "vmem", "lane", "wait", "branch", "other"]
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_") and ("_f64" in op):
        return "f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_setprio"):
        return "wait"
    if op.startswith(("s_branch", "s_cbranch")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--blocks", action="store_true", help="list every basic block")
    ap.add_argument("--trip", action="append", default=[], help="LABEL=N: trip count of the loop headed by LABEL")
    ap.add_argument("--ops", default=None, help="print the opcode histogram of the blocks whose label matches this regex")
    args = ap.parse_args()
    lines = open(args.asm).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.endswith(":") or "; @" in l:
            name = l.split(":")[0]
            if args.kernel in name and not name.startswith(("\t", ".")):
                start = i
                break
    if start is None:
        sys.exit(f"kernel matching {args.kernel!r} not found")
    blocks = []  # (label, Counter, marks, [ops], branch targets)
    cur = ["<entry>", collections.Counter(), [], [], []]
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            blocks.append(cur)
            cur = [m.group(1), collections.Counter(), [], [], []]
            continue
        if "PCMARK" in s:
            cur[2].append(s.split("PCMARK", 1)[1].strip())
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = classify(op)
        cur[1][c] += 1
        cur[3].append(op)
        if c == "branch":
            t = s.split()[-1]
            cur[4].append(t)
    blocks.append(cur)
    order = {b[0]: i for i, b in enumerate(blocks)}
    heads = set()
    for i, b in enumerate(blocks):
        for t in b[4]:
            if t in order and order[t] <= i:
                heads.add(t)
    total = collections.Counter()
    for b in blocks:
        total.update(b[1])
    print("kernel:", lines[start].split(":")[0])
    print("static totals:", {k: total[k] for k in CLASSES if total[k]})
    if args.blocks:
        print(f"{'block':14s} " + " ".join(f"{c:>6s}" for c in CLASSES) + "  notes")
        for b in blocks:
            n = sum(b[1].values())
            if n == 0 and not b[2]:
                continue
            notes = []
            if b[0] in heads:
                notes.append("LOOP-HEAD")
            for t in b[4]:
                if t in order and order[t] <= order[b[0]]:
                    notes.append(f"back->{t}")
            notes += [f"[{m}]" for m in b[2]]
            print(f"{b[0]:14s} " + " ".join(f"{b[1][c]:6d}" for c in CLASSES) + "  " + " ".join(notes))
    if args.ops:
        h = collections.Counter()
        for b in blocks:
            if re.search(args.ops, b[0]):
                h.update(b[3])
        for op, n in h.most_common():
            print(f"{n:6d} {op}")


if __name__ == "__main__":
    main()
