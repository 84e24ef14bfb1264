#!/usr/bin/env python
"""Per-basic-block instruction histogram of one kernel in a hipcc -S listing (issue-slot accounting of a loop body).
    python tools/asm_hist.py /tmp/asm/fused.s zdce_k67_conv6_conv7_curve_bf16 [min_instr]
Categories: mfma, valu (v_* except mfma / trans), trans (v_exp/rcp/...), ds_read, ds_write, lds_dma (buffer_load ... lds),
vmem_ld, vmem_st, salu, smem, wait (s_waitcnt / s_nop), barrier, branch."""
import re
import sys
from collections import Counter, OrderedDict


def cat(ins, line):
    if ins.startswith("v_mfma"):
        return "mfma"
    if ins.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")):
        return "trans"
    if ins.startswith("v_"):
        return "valu"
    if ins.startswith("ds_read") or ins.startswith("ds_load"):
        return "ds_read"
    if ins.startswith("ds_"):
        return "ds_write"
    if ins.startswith(("buffer_load", "global_load")):
        return "lds_dma" if re.search(r"\blds\b", line) else "vmem_ld"
    if ins.startswith(("buffer_store", "global_store", "buffer_atomic", "global_atomic")):
        return "vmem_st"
    if ins.startswith("s_barrier"):
        return "barrier"
    if ins.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if ins.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if ins.startswith(("s_load", "s_buffer_load", "s_memtime")):
        return "smem"
    if ins.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    min_instr = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(rf"^\S*{re.escape(kern)}\S*:\s*", l) and not l.startswith("."))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = Counter()
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        m = re.match(r"^(\.LBB\S+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = Counter()
            continue
        if not s or s.startswith((";", "//", ".")):
            continue
        ins = s.split()[0]
        blocks[cur][cat(ins, s)] += 1
    cols = ["mfma", "valu", "trans", "ds_read", "ds_write", "lds_dma", "vmem_ld", "vmem_st", "salu", "smem", "wait", "barrier", "branch"]
    print(f"{'block':14s} " + " ".join(f"{c:>7s}" for c in cols) + "   total")
    tot = Counter()
    for b, c in blocks.items():
        n = sum(c.values())
        tot.update(c)
        if n >= min_instr:
            print(f"{b:14s} " + " ".join(f"{c.get(k, 0):7d}" for k in cols) + f"   {n}")
    print(f"{'ALL':14s} " + " ".join(f"{tot.get(k, 0):7d}" for k in cols) + f"   {sum(tot.values())}")


if __name__ == "__main__":
    main()
