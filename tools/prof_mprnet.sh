cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_mpr; mkdir -p $O
cat > $O/run.py <<'PY'
import sys; sys.path.insert(0, sys.argv[1])
import torch, mon_amd
m = mon_amd.MODELS.build(name="mprnet", seed=1, compute_dtype="bf16").to("cuda")
x = torch.rand(2,3,1080,1920,device="cuda")
for _ in range(3): m.forward({"image": x})
torch.cuda.synchronize()
PY
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $O/run.py $R > $O/run.log 2>&1)
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.reader(open(sys.argv[1])))
for r in rows[:14]:
    print(r[0][:80], r[1:5])
PY
rm -rf $O/kt
