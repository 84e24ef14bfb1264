"""In-kernel cycle stamps of K67X (build: tools/exp_build.sh ie_zdce_x.hip stamp -DIE_STAMP; run: IE_HIP_LIB=.../libie_stamp.so IE_HIP_LIB_ABI_OK=1 python tools/stamps_x.py [frames]).
Per wave and step (s_memtime cycles, 100 MHz-independent: the shader clock counter): wait + barrier | MFMA phase | tail, and three fine buckets."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, mon_amd
from mon_amd import capi
from oracle import zero_dce as o
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = mon_amd.MODELS.build(name="zero_dce_re", weights=dict(o.init_state_dict(seed=1, gain=3.0)), compute_dtype="bf16").to("cuda")
x = torch.rand(N, 3, 1080, 1920, device="cuda"); out = {"enhanced": torch.empty_like(x)}
buf = torch.zeros(3 << 20, dtype=torch.int64, device="cuda")
L = capi.lib()
m.enhance(x, out=out); torch.cuda.synchronize()
L.ie_debug_set_stamps(C.c_void_p(buf.data_ptr()))
m.enhance(x, out=out); torch.cuda.synchronize()
d = buf.cpu().reshape(3, -1, 8, 8)[2]
nb = int((d[:, 0, 3] > 0).sum())
d = d[:nb].double()
steps = d[:, :, 3].mean()
names = {0: ("stores+conv1", "dma-issue", "convert", "mfma x2 row"), 1: ("u8 stores", "mfma x1 row", "value reads", "-")}
for wv in range(8):
    w, k, t = (float(d[:, wv, i].mean() / steps) for i in range(3))
    f = [float(d[:, wv, 4 + e].mean() / steps) for e in range(4)]
    nm = names[wv >> 2]
    print(f"K67X wave{wv} ({'A' if wv < 4 else 'B'}): per step wait+barrier {w:6.0f}  mfma phase {k:6.0f}  rest {t:6.0f}  |  " +
          "  ".join(f"{nm[e]} {f[e]:5.0f}" for e in range(4)) + f"  total {w + k + t + sum(f):6.0f}  (blocks {nb}, steps {steps:.0f})")
