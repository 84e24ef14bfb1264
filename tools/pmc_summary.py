#!/usr/bin/env python
"""Per-kernel means of a rocprofv3 --pmc counter_collection.csv: MFMA pipe busy fraction, clock, LDS conflicts, waits."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    full = row["Kernel_Name"].replace("(anonymous namespace)::", "")
    tag = "mconv3" if "mconv3" in full else ("mconv" if "mconv" in full else full.split("(")[0][-40:])
    name = tag + " grid=" + row["Grid_Size"]
    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    acc[name]["_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
for name, c in acc.items():
    if "mconv" not in name and "zdce" not in name and (len(sys.argv) < 3):
        continue
    m = {k: sum(v) / len(v) for k, v in c.items()}
    out = {"kernel": name, "launches": len(c["_ns"]) // max(1, len(c) - 1), "avg_us": round(m["_ns"] / 1e3, 1)}
    if "GRBM_GUI_ACTIVE" in m:
        out["clock_GHz"] = round(m["GRBM_GUI_ACTIVE"] / 8 / m["_ns"], 3)      # counter is summed over the 8 XCDs
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:     # summed over 256 CUs x 4 SIMDs, in cycles
            out["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 256 * 4), 3)
    if "FETCH_SIZE" in m:        # KB per launch (rocprofv3 derived metrics); FETCH_SIZE x2 = the gfx950 correction for wide streaming reads
        out["fetch_MB_x2"] = round(m["FETCH_SIZE"] * 2 / 1024, 1)
    if "WRITE_SIZE" in m:
        out["write_MB"] = round(m["WRITE_SIZE"] / 1024, 1)
    for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE",
              "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_VMEM", "SQ_INSTS_SALU"):
        if k in m:
            out[k] = "%.3e" % m[k]
    print(out)
