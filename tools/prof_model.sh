cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_unet32; mkdir -p $O
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python tools/bench_models.py bf16 ${1:-unet32} > $O/run.log 2>&1)
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.reader(open(sys.argv[1])))
for r in rows[:14]:
    print(r[0][:70], r[1:5])
PY
cp "$f" $O/kernel_stats_${1:-unet32}.csv; rm -rf $O/kt
