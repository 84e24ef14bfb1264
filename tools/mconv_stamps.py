"""In-kernel timeline of the persistent 3x3 matrix-core conv (diagnostic build -DIE_MSTAMP, loaded via IE_HIP_LIB)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mon_amd import ops, capi
CL = torch.channels_last
L = capi.lib()
n, h, w = (int(sys.argv[1]) if len(sys.argv) > 1 else 4), 1080, 1920
cs = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "64").split("+")]
co = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dt = torch.bfloat16
xs = [torch.randn(n, c, h, w, device="cuda", dtype=dt).contiguous(memory_format=CL) for c in cs]
mw = ops.MConvWeights(torch.randn(co, sum(cs), 3, 3) / (sum(cs) * 9) ** 0.5, torch.zeros(co), cs, dt, "cuda")
for _ in range(3):
    ops.mconv2d(xs, mw, act="relu")
st = torch.zeros(512 * 64, dtype=torch.int64, device="cuda")
L.ie_debug_mconv_stamps.argtypes = [C.c_void_p]
L.ie_debug_mconv_stamps(C.c_void_p(st.data_ptr()))
ops.mconv2d(xs, mw, act="relu")
torch.cuda.synchronize()
L.ie_debug_mconv_stamps(None)
s = st.cpu().view(512, 64).double()
names = ["top", "vmcnt", "barrier", "dma-issue", "kx0", "kx1", "kx2"]
nch = sum((c + 31) // 32 for c in cs)
for ch in range(min(nch, 6)):
    base = ch * 8
    ok = s[:, base + 6] > 0
    print(f"chunk {ch}: " + "  ".join(f"{names[i]}+{float((s[ok, base + i] - s[ok, base + i - 1]).mean()):.0f}" for i in range(1, 7)),
          f" | total {float((s[ok, base + 6] - s[ok, base]).mean()):.0f} cycles")
ok = s[:, 48] > 0
last = (min(nch, 6) - 1) * 8 + 6
print("epilogue of first tile:", float((s[ok, 48] - s[ok, last]).mean()), "| first tile total:", float((s[ok, 48] - s[ok, 0]).mean()))
