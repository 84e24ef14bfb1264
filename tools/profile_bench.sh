#!/bin/bash
# rocprofv3 evidence for bench.py's roofline fields (run on the GPU box; writes gpurun_out/prof_<tag>/, copy the summaries to profiles/).
#   1. --kernel-trace --stats of the bench command            -> kernel_stats.csv (avg duration per kernel must agree with bench.py's)
#   2. PMC passes FETCH_SIZE | WRITE_SIZE (one counter per pass: together they exceed the hardware; 8-frame group launches) -> hbm traffic per launch (FETCH_SIZE x2: gfx950 correction, MI355X_MICROARCH.md)
#   3. PMC passes MFMA busy / clock, LDS bank conflicts, wave-cycle split
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
ARGS="--steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-board"      # (no child processes under the profiler)
(cd $R && timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py $ARGS > $O/bench_under_rocprof.json.log 2> $O/kt.err) || echo "kernel-trace pass failed"
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $R && timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace -d $O/p$i -o p$i --output-format csv -- python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-board > $O/p$i.log 2>&1) || echo "pmc pass $i failed"
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f all > $O/pmc_pass$i.txt 2>&1
  rm -rf $O/p$i
done
rm -rf $O/kt
cat $O/pmc_pass1.txt $O/pmc_pass2.txt > $O/pmc_hbm.txt; python $R/tools/traffic_from_pmc.py $O/pmc_hbm.txt > $O/traffic.json 2> $O/traffic.err || true
tail -n 5 $O/bench_under_rocprof.json.log | cut -c1-400
head -8 $O/kernel_stats.csv
cat $O/pmc_pass*.txt | cut -c1-500
