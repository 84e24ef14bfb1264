#!/usr/bin/env python
"""Instruction counts of a kernel between consecutive s_barrier instructions (= per step of the strip kernels' unrolled loops).
    python tools/asm_steps.py listing.s kernel_name [more kernel names]
Prints, per segment, (salu, branch, valu, trans, mfma, ds_read, ds_write, lds_dma, vmem_st, wait)."""
import os
import re
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_hist import cat  # noqa: E402

COLS = ("salu", "branch", "valu", "trans", "mfma", "ds_read", "ds_write", "lds_dma", "vmem_ld", "vmem_st", "wait")


def steps(lines, kern):
    start = next(i for i, l in enumerate(lines) if re.match(rf"^\S*{re.escape(kern)}\S*:\s*", l) and not l.startswith("."))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    seg, segs = Counter(), []
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        if not s or s.startswith((";", "//", ".")) or re.match(r"^\.?L\S*:", s):
            continue
        c = cat(s.split()[0], s)
        seg[c] += 1
        if c == "barrier":
            segs.append(seg)
            seg = Counter()
    segs.append(seg)
    return segs


if __name__ == "__main__":
    lines = open(sys.argv[1]).read().splitlines()
    for kern in sys.argv[2:]:
        print(kern)
        print("   seg " + " ".join(f"{c:>8s}" for c in COLS))
        for i, s in enumerate(steps(lines, kern)):
            print(f"   {i:3d} " + " ".join(f"{s.get(c, 0):8d}" for c in COLS))
