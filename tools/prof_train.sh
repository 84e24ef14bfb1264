# tools/prof_train.sh [bench_train.py args]  -- kernel-time breakdown of one training configuration (runs on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_train; mkdir -p $O
(cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python tools/bench_train.py --dtype bf16 --steps 3 "$@" > $O/run.log 2>&1)
tail -1 $O/run.log
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.reader(open(sys.argv[1])))
for r in rows[:26]:
    print(r[0].replace("(anonymous namespace)::","")[:90], r[1:5])
PY
cp $f $O/kernel_stats.csv; rm -rf $O/kt
