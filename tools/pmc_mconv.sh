#!/bin/bash
# MFMA-pipe busy fraction and clock of the matrix-core conv kernels at the U-Net layer shapes (tools/bench_mconv.py under rocprofv3 --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_mconv; mkdir -p $O
(cd $R && timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/p -o p --output-format csv -- python tools/bench_mconv.py bf16 "->" > $O/run.log 2>&1) || echo "pmc pass failed"
f=$(find $O/p -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python $R/tools/pmc_summary.py $f all | grep -E "sconv3|mconv" | cut -c1-260 > $O/pmc_mconv_r02.txt
rm -rf $O/p
cat $O/pmc_mconv_r02.txt
