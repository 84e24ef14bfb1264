import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, mon_amd
from mon_amd import capi
from oracle import zero_dce as o
N=8
m = mon_amd.MODELS.build(name="zero_dce_re", weights=dict(o.init_state_dict(seed=1, gain=3.0)), compute_dtype="bf16").to("cuda")
x = torch.rand(N, 3, 1080, 1920, device="cuda"); out = {"enhanced": torch.empty_like(x)}
buf = torch.zeros(3 << 20, dtype=torch.int64, device="cuda")
L = capi.lib()
m.enhance(x, out=out); torch.cuda.synchronize()
L.ie_debug_set_stamps(C.c_void_p(buf.data_ptr()))
m.enhance(x, out=out); torch.cuda.synchronize()
b = buf.cpu().reshape(3, -1, 8, 8)
for ki, name in enumerate(["K12", "K345", "K67"]):
    d = b[ki]
    nb = int((d[:, 0, 3] > 0).sum())
    d = d[:nb].double()
    steps = d[:, :, 3].mean()
    for wv in range(8):
        w, i, k = d[:, wv, 0].mean() / steps, d[:, wv, 1].mean() / steps, d[:, wv, 2].mean() / steps
        print(f"{name} wave{wv} ({'A' if wv < 4 else 'B'}): per step wait+barrier {w:7.0f}  dma-issue {i:6.0f}  work {k:7.0f}  total {w+i+k:7.0f}  (blocks {nb}, steps {steps:.0f})  fine " + " ".join(f"{float(d[:, wv, 4 + e].mean() / steps):6.0f}" for e in range(4)))
