# rocprofv3 kernel stats of U-Net-32 at batch 32 (tools/bench_unet32_batch.py): the bench configuration of BASELINE config 3
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_unet32_b32; rm -rf $O; mkdir -p $O
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python tools/bench_unet32_batch.py 32 ${1:-f16} > $O/run.log 2>&1)
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out/profiles_r03
cp "$f" $R/gpurun_out/profiles_r03/kernel_stats_model_unet32_${1:-f16}_batch32.csv
tail -1 $O/run.log
python - "$f" <<'PY'
import csv,sys
rows=list(csv.reader(open(sys.argv[1])))
tot=sum(float(r[2]) for r in rows[1:])
for r in rows[1:13]:
    print("   %-78s calls %4s avg %9.0f us  %5.1f%%" % (r[0][:78], r[1], float(r[3])/1e3, 100*float(r[2])/tot))
PY
rm -rf $O/kt
