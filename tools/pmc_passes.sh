cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_r04}
EXTRA="${2:-}"      # extra bench.py arguments (e.g. "--variant 256")
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $C --kernel-trace -d $O/p$i -o p$i --output-format csv -- python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-board $EXTRA > $O/p$i.log 2>&1) || echo "pass $i failed"
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f > $O/p$i.summary.txt 2>&1 && rm -rf $O/p$i
  cat $O/p$i.summary.txt 2>/dev/null | cut -c1-600
done
