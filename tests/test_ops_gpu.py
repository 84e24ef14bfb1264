"""GPU parity of the per-op C-ABI entry points against torch CPU / the plain-C oracle."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mon_amd import capi, image, ops
from oracle import c_oracle
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _conv_gpu(x, w, b, stride=1, pad=1, pad_mode="zeros", groups=1, act="none", slope=0.0, x1=None):
    """x, x1: NCHW cpu tensors; runs ie_conv2d_nhwc_f32 and returns NCHW cpu."""
    L = capi.lib()
    st = capi.current_stream()
    n, c0, h, wd = x.shape
    xs = x.permute(0, 2, 3, 1).contiguous().cuda()
    x1s = None if x1 is None else x1.permute(0, 2, 3, 1).contiguous().cuda()
    co, cig, kh, kw = w.shape
    wdev = torch.empty(kh * kw * cig * co, device="cuda")
    wc = w.contiguous()
    capi.check(L.ie_pack_weight_hwio_f32(capi.ptr(wc), co, cig, kh, kw, capi.ptr(wdev), st))
    bdev = None if b is None else b.cuda()
    d = capi.Conv2dDesc(n, h, wd, c0, 0 if x1 is None else x1.shape[1], co, kh, kw, stride, pad, capi.PAD[pad_mode],
                        groups, capi.ACT[act], slope, 0, 0)
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (wd + 2 * pad - kw) // stride + 1
    out = torch.empty((n, ho, wo, co), device="cuda")
    capi.check(L.ie_conv2d_nhwc_f32(C.byref(d), capi.ptr(xs), capi.ptr(x1s), capi.ptr(wdev), capi.ptr(bdev), capi.ptr(out), st),
               "ie_conv2d_nhwc_f32")
    assert (d.Ho, d.Wo) == (ho, wo)
    return out.permute(0, 3, 1, 2).contiguous().cpu()


CASES = [
    dict(ci=32, co=32, k=3, s=1, p=1, mode="zeros", act="relu"),          # zero_dce.py:134
    dict(ci=3, co=32, k=3, s=1, p=1, mode="zeros", act="relu"),           # first layer, Cin % 4 != 0
    dict(ci=4, co=64, k=9, s=1, p=4, mode="replicate", act="none"),       # retinexnet/model.py:26
    dict(ci=64, co=64, k=3, s=2, p=1, mode="replicate", act="relu"),      # retinexnet/model.py:60
    dict(ci=192, co=64, k=1, s=1, p=1, mode="replicate", act="none"),     # retinexnet/model.py:66 (grows the map)
    dict(ci=64, co=1, k=3, s=1, p=0, mode="zeros", act="none"),           # retinexnet/model.py:67
    dict(ci=64, co=64, k=4, s=2, p=1, mode="zeros", act="none", bias=False),  # hinet.py:65
    dict(ci=32, co=24, k=3, s=1, p=1, mode="zeros", act="tanh"),
    dict(ci=16, co=3, k=3, s=1, p=1, mode="zeros", act="sigmoid"),        # zero_ig.py:331
    dict(ci=32, co=30, k=1, s=1, p=0, mode="zeros", act="lrelu", slope=0.2),
]


@pytest.mark.parametrize("c", CASES, ids=lambda c: f"{c['ci']}-{c['co']}-k{c['k']}s{c['s']}p{c['p']}{c['mode'][0]}-{c['act']}")
def test_conv2d_generic(c):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, c["ci"], 21, 37, generator=g)
    w = torch.randn(c["co"], c["ci"], c["k"], c["k"], generator=g) / (c["ci"] * c["k"] ** 2) ** 0.5
    b = None if c.get("bias") is False else torch.randn(c["co"], generator=g)
    xp = F.pad(x, (c["p"],) * 4, mode="replicate") if c["mode"] == "replicate" else x
    ref = F.conv2d(xp, w, b, stride=c["s"], padding=0 if c["mode"] == "replicate" else c["p"])
    ref = {"none": lambda t: t, "relu": F.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid,
           "lrelu": lambda t: F.leaky_relu(t, c.get("slope", 0.2))}[c["act"]](ref)
    got = _conv_gpu(x, w, b, c["s"], c["p"], c["mode"], 1, c["act"], c.get("slope", 0.0))
    assert got.shape == ref.shape
    assert rel_err(got, ref) <= 2e-5


def test_conv2d_two_sources_is_channel_concat():
    g = torch.Generator().manual_seed(2)
    a, b2 = torch.randn(1, 32, 19, 23, generator=g), torch.randn(1, 32, 19, 23, generator=g)
    w = torch.randn(32, 64, 3, 3, generator=g) * 0.05
    bias = torch.randn(32, generator=g)
    ref = F.relu(F.conv2d(torch.cat([a, b2], 1), w, bias, padding=1))
    got = _conv_gpu(a, w, bias, act="relu", x1=b2)
    assert rel_err(got, ref) <= 2e-5


def test_conv2d_depthwise_matches_c_oracle():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 15, 18, generator=g)
    w = torch.randn(32, 1, 3, 3, generator=g) * 0.3
    b = torch.randn(32, generator=g)
    ref = c_oracle.conv2d(x.numpy(), w.numpy(), b.numpy(), groups=32)
    got = _conv_gpu(x, w, b, groups=32)
    assert rel_err(got, ref) <= 2e-5


def test_conv2d_rejects_bad_arguments():
    L = capi.lib()
    d = capi.Conv2dDesc(1, 8, 8, 4, 0, 4, 3, 3, 1, 1, 0, 2, 0, 0.0, 0, 0)     # groups=2 unsupported
    t = torch.zeros(1024, device="cuda")
    assert L.ie_conv2d_nhwc_f32(C.byref(d), capi.ptr(t), None, capi.ptr(t), None, capi.ptr(t), None) == -3
    assert b"groups" in L.ie_last_error()
    d = capi.Conv2dDesc(1, 8, 8, 0, 0, 4, 3, 3, 1, 1, 0, 1, 0, 0.0, 0, 0)
    assert L.ie_conv2d_nhwc_f32(C.byref(d), capi.ptr(t), None, capi.ptr(t), None, capi.ptr(t), None) == -1


@pytest.mark.parametrize("dt,tol", [("f32", 0.0), ("f16", 1e-3), ("bf16", 8e-3)])
def test_layout_roundtrip(dt, tol):
    L = capi.lib()
    st = capi.current_stream()
    x = torch.rand(2, 3, 11, 13, generator=torch.Generator().manual_seed(4)).cuda()
    esz = {"f32": 4, "f16": 2, "bf16": 2}[dt]
    mid = torch.zeros(2 * 11 * 13 * 8 * esz, dtype=torch.uint8, device="cuda")
    back = torch.empty_like(x)
    capi.check(L.ie_nchw_f32_to_nhwc(capi.ptr(x), capi.ptr(mid), capi.DTYPES[dt], 2, 3, 11, 13, 8, st))
    capi.check(L.ie_nhwc_to_nchw_f32(capi.ptr(mid), capi.DTYPES[dt], capi.ptr(back), 2, 3, 11, 13, 8, st))
    assert float((back - x).abs().max()) <= tol
    if dt == "f32":
        nhwc = mid.view(torch.float32).reshape(2, 11, 13, 8).cpu()
        assert torch.equal(nhwc[..., :3], x.cpu().permute(0, 2, 3, 1)) and float(nhwc[..., 3:].abs().max()) == 0.0


def test_uint8_rules_and_back():
    x = torch.cat([torch.rand(1, 3, 31, 33, generator=torch.Generator().manual_seed(5)) * 1.2 - 0.1,
                   torch.tensor([0.0, 1.0, 0.5, 0.49803922, 127.5 / 255, 254.5 / 255, -3.0, 7.0, 1 / 255, 0.00196]).repeat(3, 1)
                   .reshape(1, 3, 1, 10).expand(1, 3, 31, 10)], dim=3).contiguous()
    ref_round = x.mul(255).add(0.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
    ref_trunc = x.mul(255).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
    assert torch.equal(image.to_uint8(x.cuda(), "save_image").cpu(), ref_round)
    assert torch.equal(image.to_uint8(x.cuda(), "truncate").cpu(), ref_trunc)
    u = torch.randint(0, 256, (2, 9, 7, 3), dtype=torch.uint8)
    assert torch.equal(image.to_float(u.cuda()).cpu(), (u.float() / 255.0).permute(0, 3, 1, 2))
    # u8 -> float -> u8 is the identity under the save_image rule
    assert torch.equal(image.to_uint8(image.to_float(u.cuda())).cpu(), u)


@pytest.mark.parametrize("ac", [0, 1])
@pytest.mark.parametrize("sizes", [((40, 72), (64, 96)), ((64, 96), (40, 72)), ((1080 // 8, 1920 // 8), (1088 // 8, 1920 // 8)), ((5, 7), (10, 14))])
def test_resize_bilinear(ac, sizes):
    (hi, wi), (ho, wo) = sizes
    x = torch.rand(2, 3, hi, wi, generator=torch.Generator().manual_seed(6))
    ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=bool(ac))
    out = torch.empty(2, 3, ho, wo, device="cuda")
    capi.check(capi.lib().ie_resize_bilinear_nchw_f32(capi.ptr(x.cuda()), capi.ptr(out), 2, 3, hi, wi, ho, wo, ac, capi.current_stream()))
    assert float((out.cpu() - ref).abs().max()) <= 2e-6


def test_le_curve_matches_reference_order_bitwise():
    g = torch.Generator().manual_seed(7)
    x = torch.rand(2, 3, 17, 19, generator=g)
    r = torch.rand(2, 24, 17, 19, generator=g) * 2 - 1
    from oracle import zero_dce as o
    ref = o.curve(x, r)
    out = torch.empty(2, 3, 17, 19, device="cuda")
    adj = r.permute(0, 2, 3, 1).contiguous().cuda()
    capi.check(capi.lib().ie_le_curve(capi.ptr(x.cuda()), capi.ptr(adj), 24, capi.ptr(out), 2, 17, 19, 8, capi.current_stream()))
    assert torch.equal(out.cpu(), ref)    # same fp32 association order, no FMA contraction


@pytest.mark.parametrize("co,cs", [(64, [32]), (136, [24, 16])], ids=["tile8x4", "tile8x8-ragged"])
def test_conv_f32_large_tiles(co, cs):
    """>= 128K output pixels select the 8x4 / 8x8 register tiles of the LDS-tiled fp32 conv (small cases use 4x4)."""
    from mon_amd import ops
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(2, c, 260, 270, generator=g) for c in cs]
    w = torch.randn(co, sum(cs), 3, 3, generator=g) / (9 * sum(cs)) ** 0.5
    b = torch.randn(co, generator=g)
    ref = F.relu(F.conv2d(torch.cat(xs, 1), w, b, padding=1))
    out = ops.conv2d([ops.to_nhwc(x.cuda()) for x in xs], ops.ConvWeights(w, b, "cuda"), act="relu")
    assert rel_err(ops.to_nchw(out).cpu(), ref) <= 2e-5


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_retinex_single_pass_ends(dt):
    """ie_retinex_decom_input == cat(max_c(x), x) rounded once to 16 bits, padded to 8 channels (retinexnet/model.py:45-46);
    ie_retinex_outputs_f32 == (R, L, R * I_delta) as NCHW fp32, bit for bit (model.py:72-74, 229-231)."""
    tdt = ops.TORCH_DT[dt]
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 19, 37, generator=g)
    y = ops.retinex_decom_input(x.cuda(), tdt)
    want = torch.cat([x.max(1, keepdim=True)[0], x, torch.zeros(2, 4, 19, 37)], 1).to(tdt)
    assert tuple(y.shape) == (2, 8, 19, 37) and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y.cpu().float(), want.float())
    rl = torch.rand(2, 4, 19, 37, generator=g)
    idl = torch.rand(2, 1, 19, 37, generator=g) * 3
    R, L, S = ops.retinex_outputs(ops.to_nhwc(rl.cuda()), ops.to_nhwc(idl.cuda()))
    assert R.is_contiguous() and L.is_contiguous() and S.is_contiguous()
    assert torch.equal(R.cpu(), rl[:, :3]) and torch.equal(L.cpu(), rl[:, 3:]) and torch.equal(S.cpu(), rl[:, :3] * idl)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_16bit_replicate_pad_and_zero_insert(dt):
    """ie_pad_replicate_nhwc / its adjoint / ie_zero_insert2_nhwc on 16-bit tensors: the pieces that put RetinexNet's replicate-padded and stride-2 layers on the
    16-bit training graph.  The pad is a copy (bit-exact); its adjoint sums the copies in fp32 and rounds once; <conv_s2(x), dy> == <x, conv_s1(zero_insert(dy))>."""
    tdt = ops.TORCH_DT[dt]
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 16, 9, 13, generator=g).to(tdt)
    xn = x.cuda().contiguous(memory_format=torch.channels_last)
    y = ops.pad_replicate(xn, 3)
    assert torch.equal(y.float().cpu(), F.pad(x.float(), (3, 3, 3, 3), mode="replicate"))
    dy = torch.randn(2, 16, 15, 19, generator=g).to(tdt)
    xf = x.float().requires_grad_(True)
    F.pad(xf, (3, 3, 3, 3), mode="replicate").backward(dy.float())
    dx = torch.zeros_like(xn)
    ops.pad_replicate_backward_into(dy.cuda().contiguous(memory_format=torch.channels_last), dx, 3, False)
    assert rel_err(dx.float().cpu(), xf.grad) <= (1e-3 if dt == "f16" else 8e-3)
    base = torch.randn_like(x).cuda().contiguous(memory_format=torch.channels_last)
    dx2 = base.clone()
    ops.pad_replicate_backward_into(dy.cuda().contiguous(memory_format=torch.channels_last), dx2, 3, True)
    assert rel_err(dx2.float().cpu(), xf.grad + base.float().cpu()) <= (2e-3 if dt == "f16" else 1.6e-2)
    z = ops.zero_insert2(xn, 18, 27)
    want = torch.zeros(2, 16, 18, 27)
    want[:, :, 0:17:2, 0:25:2] = x.float()
    assert torch.equal(z.float().cpu(), want)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("mode", ["bilinear_align", "bilinear", "nearest"])
@pytest.mark.parametrize("case", [((2, 16, 7, 9), (14, 18), (0, 1, 0, 0)), ((1, 24, 5, 6), (10, 12), (1, 0, 2, 1)), ((3, 8, 4, 4), (9, 7), (0, 0, 0, 3)),
                                  ((1, 256, 67, 120), (134, 240), (0, 1, 0, 0))], ids=lambda c: "x".join(map(str, c[0])))
def test_resize_into_a_padded_tensor(case, mode, dt):
    """ie_resize_pad_nhwc (`Up.forward`: up-sample, then F.pad to the skip tensor's size, unet.py:95-103) = ie_resize_nhwc + ie_pad_nhwc, bit for bit."""
    shape, size, pads = case
    tdt = ops.TORCH_DT[dt]
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(5)).to(tdt).cuda().contiguous(memory_format=torch.channels_last)
    got = ops.resize_pad(x, size, pads, mode)
    want = ops.pad(ops.resize(x, size, mode), *pads)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)
    ref = F.pad(F.interpolate(x.float(), size=size, mode="nearest" if mode == "nearest" else "bilinear",
                              **({} if mode == "nearest" else {"align_corners": mode == "bilinear_align"})), (pads[2], pads[3], pads[0], pads[1]))
    assert float((got.float() - ref).abs().max()) <= (4e-3 if dt == "f16" else 3e-2) * max(1.0, float(ref.abs().max()))

