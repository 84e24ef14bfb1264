# PMC passes (HBM traffic, MFMA pipe busy) of one model's forward loop: bash tools/pmc_models.sh <tag> -- python tools/bench_unet32_batch.py 32 f16
# One counter group per pass, --kernel-trace only (never with sys / hip traces).  Summaries land in gpurun_out/pmc_<tag>.txt
TAG=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O
: > $R/gpurun_out/pmc_$TAG.txt
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  (cd $R && timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace -d $O/p$i -o p$i --output-format csv -- "$@" > $O/p$i.log 2>&1) || echo "pass $i failed"
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $C" >> $R/gpurun_out/pmc_$TAG.txt
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f all >> $R/gpurun_out/pmc_$TAG.txt 2>&1
  rm -rf $O/p$i
done
cut -c1-260 $R/gpurun_out/pmc_$TAG.txt
