#!/usr/bin/env python3
"""Developer tool: static instruction counts per kernel of a device assembly file (hipcc -S --cuda-device-only), optionally as a
diff against a second file.  Static counts say nothing about trip counts; they are a quick regression screen for a compiler flag
or a source change across ALL kernels of the translation unit.

    python tools/asm_counts.py base.s [other.s] [--filter REGEX]
"""
import argparse
import collections
import re
import subprocess


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("void ", "") for o in out]


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith(("v_readlane", "v_writelane")):
        return "lane"
    if op.startswith("v_") and "_f64" in op:
        return "f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    return "other"


def parse(path):
    kernels = collections.OrderedDict()
    cur = None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = kernels.setdefault(m.group(1), collections.Counter())
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is None:
            continue
        t = line.strip()
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        cur[classify(op)] += 1
        cur["total"] += 1
    return kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("base")
    ap.add_argument("other", nargs="?")
    ap.add_argument("--filter", default=".")
    a = ap.parse_args()
    kb = parse(a.base)
    ko = parse(a.other) if a.other else None
    names = list(kb)
    dn = dict(zip(names, demangle(names)))
    cols = ["total", "valu", "pk", "f64", "mfma", "lane", "salu", "smem", "lds", "vmem", "scratch", "wait", "nop"]
    print(f"{'kernel':64s} " + " ".join(f"{c:>7s}" for c in cols))
    for n in names:
        if not re.search(a.filter, dn[n]):
            continue
        row = kb[n]
        if ko is None:
            print(f"{dn[n][:64]:64s} " + " ".join(f"{row[c]:7d}" for c in cols))
        elif n in ko:
            d = {c: ko[n][c] - row[c] for c in cols}
            if any(d.values()):
                print(f"{dn[n][:64]:64s} " + " ".join(f"{row[c]:7d}" if c == "total" else f"{d[c]:+7d}" for c in cols) + f"  (total {d['total']:+d})")


if __name__ == "__main__":
    main()
