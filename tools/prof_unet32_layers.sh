# per-dispatch durations of the last U-Net-32 forward at batch 32 (which layer is which kernel call): rocprofv3 kernel trace of tools/bench_unet32_batch.py
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_unet32_layers; rm -rf $O; mkdir -p $O
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- python tools/bench_unet32_batch.py 32 ${1:-f16} > $O/run.log 2>&1)
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# one forward = from one unet_first32 call to the next
starts = [i for i, n in enumerate(names) if "unet_first32" in n]
a, b = starts[-2], starts[-1]
tot = 0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print("   %-60s grid %9s wg %4s lds %6s  %8.0f us" % (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60], r["Grid_Size_X"], r["Workgroup_Size_X"], r.get("LDS_Block_Size", "?"), d))
print("   kernels of one forward: %.0f us; wall between the two first-level launches: %.0f us" % (tot, (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3))
PY
rm -rf $O/kt
