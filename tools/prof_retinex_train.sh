# rocprofv3 kernel stats of RetinexNet's decom training phase on the 16-bit graph (tools/bench_retinex_train.py 2 <graph> <phase>)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_retinex_train; rm -rf $O; mkdir -p $O
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python tools/bench_retinex_train.py 2 ${1:-bf16} ${2:-decom} > $O/run.log 2>&1)
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out/profiles_r03
cp "$f" $R/gpurun_out/profiles_r03/kernel_stats_train_retinexnet_${2:-decom}_${1:-bf16}.csv
tail -2 $O/run.log
python - "$f" <<'PY'
import csv,sys
rows=list(csv.reader(open(sys.argv[1])))
tot=sum(float(r[2]) for r in rows[1:])
for r in rows[1:16]:
    print("   %-84s calls %4s avg %9.0f us  %5.1f%%" % (r[0].replace("(anonymous namespace)::","")[:84], r[1], float(r[3])/1e3, 100*float(r[2])/tot))
PY
rm -rf $O/kt
