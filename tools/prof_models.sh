# rocprofv3 kernel stats of one forward-loop per model (tools/bench_models.py <dtype> <model>): profiles/<round>/kernel_stats_model_<model>_<dtype>.csv
# usage (on the GPU box, from the repo root): bash tools/prof_models.sh r03 f16 unet32 retinexnet unet64
RND=$1; DT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for M in "$@"; do
  O=$R/gpurun_out/prof_$M; rm -rf $O; mkdir -p $O
  (cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python tools/bench_models.py $DT $M > $O/run.log 2>&1)
  f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
  mkdir -p $R/gpurun_out/profiles_$RND
  cp "$f" $R/gpurun_out/profiles_$RND/kernel_stats_model_${M}_${DT}.csv
  tail -1 $O/run.log
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.reader(open(sys.argv[1])))
for r in rows[:12]:
    print("   ", r[0][:86], r[1:5])
PY
  rm -rf $O/kt
done
