# rocprofv3 kernel stats of one training configuration of tools/bench_train.py: bash tools/prof_train_any.sh <tag> <bench_train.py args...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_train_$TAG; rm -rf $O; mkdir -p $O
(cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python tools/bench_train.py "$@" > $O/run.log 2>&1)
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out/profiles_r03
cp "$f" $R/gpurun_out/profiles_r03/kernel_stats_train_$TAG.csv
tail -1 $O/run.log
python - "$f" <<'PY'
import csv,sys
rows=list(csv.reader(open(sys.argv[1])))
tot=sum(float(r[2]) for r in rows[1:])
for r in rows[1:14]:
    print("   %-84s calls %4s avg %9.0f us  %5.1f%%" % (r[0].replace("(anonymous namespace)::","")[:84], r[1], float(r[3])/1e3, 100*float(r[2])/tot))
PY
rm -rf $O/kt
