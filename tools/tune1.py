import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, mon_amd
from mon_amd import capi
from oracle import zero_dce as o
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
m = mon_amd.MODELS.build(name="zero_dce_re", weights=dict(o.init_state_dict(seed=1, gain=3.0)), compute_dtype=dtype, variant=int(os.environ.get("VAR", "0"))).to("cuda")
x = torch.rand(N, 3, 1080, 1920, device="cuda"); out = {"enhanced": torch.empty_like(x)}
L = capi.lib()
for rep in range(2):
    for _ in range(2): m.enhance(x, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m.enhance(x, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    ms = (C.c_float * 8)(); names = (C.c_char_p * 8)(); launches = (C.c_int * 8)(); cnt = C.c_int(0)
    capi.check(L.ie_zdce_profile_forward(m._ensure_plan(), capi.ptr(x), capi.ptr(out["enhanced"]), None, None, N, 1080, 1920, capi.current_stream(), ms, names, launches, 8, C.byref(cnt)))
    print(f"{dt*1e3:7.2f} ms {N/dt:7.1f} img/s | " + " ".join(f"{ms[i]:.2f}" for i in range(7)), flush=True)
