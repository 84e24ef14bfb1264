"""GPU parity of the 16-bit matrix-core convolution (ie_mconv_nhwc) against torch CPU fp32 on the SAME 16-bit-rounded
inputs and weights: the only differences left are fp32 accumulation order and the rounding of the stored result."""
import pytest
import torch
import torch.nn.functional as F

from mon_amd import capi, ops
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
CL = torch.channels_last

CASES = [
    dict(cs=[32], co=32, k=3, p=1),
    dict(cs=[64], co=64, k=3, p=1, act="relu"),                       # unet.py:43-48 (BN folded)
    dict(cs=[8], co=32, k=3, p=1, act="relu"),                        # first layer: RGB padded to 8 channels
    dict(cs=[64, 64], co=64, k=3, p=1, act="relu"),                   # decoder concat [skip, up] (unet.py:97)
    dict(cs=[32, 32, 32, 32, 64], co=32, k=3, p=1, act="lrelu"),      # LLUnet++ conv0_4 (llunetpp.py:239)
    dict(cs=[40, 24], co=72, k=3, p=1, bias=False),                   # short chunks, ragged Cout tile
    dict(cs=[256], co=128, k=3, p=1, act="relu"),                     # two cout tiles, 8 K chunks
    dict(cs=[32], co=3, k=1, p=0, clamp=(0.0, 1.0), out_f32=True),    # outc + clamp, fp32 result (llunetpp.py:241)
    dict(cs=[64], co=24, k=3, p=1, act="tanh"),
    dict(cs=[8], co=64, k=9, p=4, mode="replicate"),                  # retinexnet/model.py:26 (4 -> padded 8)
    dict(cs=[64], co=64, k=3, p=1, s=2, mode="replicate", act="relu"),  # retinexnet/model.py:60
    dict(cs=[192], co=64, k=1, p=1, mode="replicate"),                # retinexnet/model.py:66 (grows the map)
    dict(cs=[64], co=1, k=3, p=0),                                    # retinexnet/model.py:67
    dict(cs=[32], co=32, k=4, p=1, s=2, bias=False),                  # hinet.py:65
    dict(cs=[64], co=64, k=3, p=1, act="lrelu", residual=True),       # hinet.py:77 (+ identity)
    dict(cs=[16], co=3, k=3, p=1, act="sigmoid"),                     # zero_ig.py:331
]


def _id(c):
    return f"{'+'.join(map(str, c['cs']))}-{c['co']}-k{c['k']}s{c.get('s', 1)}p{c['p']}{c.get('mode', 'zeros')[0]}-{c.get('act', 'none')}"


# The narrow 3x3 layers leave ie_mconv_nhwc through the strip kernel (ie_conv_strip.hip) once the image is at least 64 columns wide:
# two strips with a ragged second one, one and several row chunks, every channel / activation / padding combination it covers.
STRIP_CASES = [
    dict(cs=[32], co=32, k=3, p=1),
    dict(cs=[32], co=32, k=3, p=1, act="relu", bias=False),
    dict(cs=[64], co=64, k=3, p=1, act="relu"),                       # unet.py:43-48 at full / half resolution
    dict(cs=[64], co=64, k=3, p=1, act="relu", mode="replicate"),     # retinexnet/model.py:28-40
    dict(cs=[32, 32], co=32, k=3, p=1, act="lrelu"),                  # decoder concat [skip, up] of U-Net-32
    dict(cs=[32, 32], co=64, k=3, p=1),
    dict(cs=[64], co=32, k=3, p=1, act="lrelu", mode="replicate"),
    dict(cs=[32], co=64, k=3, p=1, act="relu"),                       # down1 of U-Net-32
    dict(cs=[64, 64], co=64, k=3, p=1, act="relu"),                   # decoder concat [skip, up] of U-Net-64: four K chunks
    dict(cs=[128], co=64, k=3, p=1),
    dict(cs=[64, 64], co=64, k=3, p=1, act="relu", mode="replicate"), # retinexnet/model.py:62-64
    dict(cs=[64], co=64, k=3, p=1, s=2, act="relu", mode="replicate"),  # retinexnet/model.py:60: stride 2 = the even rows / columns of the strip kernel
    dict(cs=[32], co=32, k=3, p=1, s=2),
    dict(cs=[32, 32], co=64, k=3, p=1, s=2, act="lrelu"),
]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 37, 150), (1, 200, 64), (3, 5, 129)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("c", STRIP_CASES, ids=_id)
def test_strip_conv_vs_torch(c, dt, shape):
    test_mconv_vs_torch(c, dt, shape)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 37, 150), (1, 200, 64), (3, 6, 129), (1, 270, 480), (2, 2, 66)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("cs,co,act", [([32], 32, "relu"), ([64], 64, "relu"), ([32], 64, "relu"), ([32, 32], 32, "relu"), ([64, 64], 64, "relu"),
                                       ([64], 64, "none"), ([128], 128, "relu"), ([64], 128, "relu"), ([128], 16, "relu"), ([96], 32, "relu"), ([128], 128, "lrelu"),
                                       ([256], 80, "relu"), ([320], 64, "relu")], ids=lambda v: str(v).replace(" ", ""))
def test_conv_with_fused_maxpool(cs, co, act, dt, shape):
    """ie_mconv_pool_nhwc (unet.py:70-72, MaxPool2d(2) behind the DoubleConv): the full-resolution output is the plain conv's, bit for bit,
    and the pooled copy is F.max_pool2d of it, bit for bit -- from the strip kernel's parked rows (first five cases; odd heights and widths
    drop the last row / column like torch), from the register-weights kernel's epilogue (wide ReLU layers: row pairs are a lane's own accumulators,
    column pairs neighbouring lanes; all three wave layouts; 8-row tiles for few K chunks, 16-row tiles for [320] -> 64 at 200 / 270 rows), or from the conv
    followed by the pool kernel (no ReLU)."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(23)
    xs = [torch.randn(n, ci, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL) for ci in cs]
    cin = sum(cs)
    wt = (torch.randn(co, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(tdt).float()
    mw = ops.MConvWeights(wt, torch.randn(co, generator=g), cs, tdt, "cuda")
    plain = ops.mconv2d(xs, mw, act=act)
    out, pooled = ops.mconv2d(xs, mw, act=act, pool=True)
    assert torch.equal(out, plain)
    assert tuple(pooled.shape) == (n, co, h // 2, w // 2)
    assert torch.equal(pooled.float().cpu(), F.max_pool2d(plain.float().cpu(), 2))
    with pytest.raises(capi.IEError):
        ops.mconv2d(xs, mw, act=act, pool=True, out_f32=True)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 37, 152), (1, 200, 64), (3, 5, 132), (1, 270, 480)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("cs,co,ct,clamp", [([32], 32, 3, (0.0, 1.0)), ([64], 64, 3, None), ([32, 32], 32, 3, (0.0, 1.0)), ([64], 32, 16, None),
                                            ([64, 64], 64, 1, (-0.5, 0.5))], ids=lambda v: str(v).replace(" ", ""))
def test_conv_with_fused_1x1_tail(cs, co, ct, clamp, dt, shape):
    """ie_mconv_tail_nhwc (the U-Net head, unet.py:186-195): 3x3 ReLU conv -> 1x1 conv -> clamp -> NCHW fp32 in one launch equals the two
    ie_mconv_nhwc calls (same 16-bit rounding of the intermediate, same matrix-core contraction; fp32 summation order may differ)."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(29)
    xs = [torch.randn(n, ci, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL) for ci in cs]
    cin = sum(cs)
    wt = (torch.randn(co, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(tdt).float()
    mw = ops.MConvWeights(wt, torch.randn(co, generator=g), cs, tdt, "cuda")
    w1 = (torch.randn(ct, co, 1, 1, generator=g) / co ** 0.5).to(tdt).float()
    b1 = torch.randn(ct, generator=g) * 0.1
    tail = ops.MTailWeights(w1, b1, tdt, "cuda")
    if cin > 64:      # four K chunks: the fused head (and pool) forms of that shape spilled registers and were retired -- callers run the two launches
        assert not ops.mconv2d_tail_applicable(xs, mw, tail)
        with pytest.raises(capi.IEError):
            ops.mconv2d_tail(xs, mw, tail, clamp=clamp)
        out, pooled = ops.mconv2d(xs, mw, act="relu", pool=True)          # ... and conv + the streaming max-pool kernel behind the same call
        assert torch.equal(out, ops.mconv2d(xs, mw, act="relu")) and torch.equal(pooled.float().cpu(), F.max_pool2d(out.float().cpu(), 2))
        return
    assert ops.mconv2d_tail_applicable(xs, mw, tail)
    got = ops.mconv2d_tail(xs, mw, tail, clamp=clamp)
    mid = ops.mconv2d(xs, mw, act="relu")
    want = ops.to_nchw(ops.mconv2d(mid, ops.MConvWeights(w1, b1, [co], tdt, "cuda"), pad=0, clamp=clamp, out_f32=True))
    assert tuple(got.shape) == (n, ct, h, w) and got.dtype == torch.float32 and got.is_contiguous()
    assert rel_err(got.cpu(), want.cpu()) <= 2e-6
    ref = F.conv2d(F.relu(F.conv2d(torch.cat([x.float().cpu() for x in xs], 1), wt, mw.b.cpu(), padding=1)).to(tdt).float(), w1, b1)
    ref = ref if clamp is None else ref.clamp(*clamp)
    assert rel_err(got.cpu(), ref) <= (3e-3 if dt == "f16" else 2e-2)       # a 1-ulp difference in the rounded intermediate is amplified by the 1x1


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("pad_mode", ["zeros", "replicate"])
@pytest.mark.parametrize("shape", [(2, 37, 150), (1, 200, 64), (3, 6, 129), (1, 270, 480), (2, 1, 66), (1, 2, 253), (1, 131, 126)], ids=lambda s: "x".join(map(str, s)))
def test_two_stacked_convs_in_one_launch(shape, pad_mode, dt):
    """ie_mconv_pair_nhwc (DecomNet's Conv3x3(64, 64) + ReLU stack, retinexnet/model.py:28-33): two layers in one launch, the intermediate in LDS
    only, equal two ie_mconv_nhwc calls bit for bit -- the same contraction order and the same single rounding of the intermediate; replicate
    padding pads the intermediate of the second layer like the reference's second Conv2d does."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(31)
    x = torch.randn(n, 64, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL)
    mws = [ops.MConvWeights((torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(tdt).float(), torch.randn(64, generator=g) * 0.2, [64], tdt, "cuda") for _ in range(2)]
    assert ops.mconv2d_pair_applicable(x, mws[0], mws[1], pad_mode=pad_mode)
    got = ops.mconv2d_pair(x, mws[0], mws[1], pad_mode=pad_mode)
    mid = ops.mconv2d(x, mws[0], pad_mode=pad_mode, act="relu")
    want = ops.mconv2d(mid, mws[1], pad_mode=pad_mode, act="relu")
    assert tuple(got.shape) == tuple(want.shape) and got.dtype == want.dtype
    assert torch.equal(got, want)
    assert not ops.mconv2d_pair_applicable(x[:, :32].contiguous(memory_format=CL), mws[0], mws[1])


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("pad_mode", ["zeros", "replicate"])
@pytest.mark.parametrize("cout,act", [(4, "sigmoid"), (8, "none"), (16, "tanh"), (12, "relu")])
@pytest.mark.parametrize("shape", [(2, 37, 150), (1, 200, 64), (3, 6, 129), (2, 1, 66), (1, 2, 253), (1, 131, 126)], ids=lambda s: "x".join(map(str, s)))
def test_stacked_conv_with_thin_second_layer(shape, cout, act, pad_mode, dt):
    """ie_mconv_pair_thin_nhwc (DecomNet's last Conv3x3(64, 64) + ReLU and `net1_recon` Conv3x3(64, 4) + sigmoid, retinexnet/model.py:33-36,52-56) in one
    launch: the fp32 output equals two ie_mconv_nhwc calls to fp32 rounding (same 16-bit intermediate, fp32 accumulation in another order), the optional
    16-bit 8-channel copy is that output rounded once."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(37)
    x = torch.randn(n, 64, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL)
    wa = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(tdt).float()
    wb = (torch.randn(cout, 64, 3, 3, generator=g) / 24.0).to(tdt).float()
    mwa = ops.MConvWeights(wa, torch.randn(64, generator=g) * 0.2, [64], tdt, "cuda")
    mwb = ops.MConvWeights(wb, torch.randn(cout, generator=g) * 0.2, [64], tdt, "cuda")
    assert ops.mconv2d_pair_thin_applicable(x, mwa, mwb, pad_mode=pad_mode, act_b=act)
    copy16 = cout == 4
    got = ops.mconv2d_pair_thin(x, mwa, mwb, pad_mode=pad_mode, act_b=act, copy16=copy16)
    got, got16 = got if copy16 else (got, None)
    mid = ops.mconv2d(x, mwa, pad_mode=pad_mode, act="relu")
    want = ops.mconv2d(mid, mwb, pad_mode=pad_mode, act=act, out_f32=True)
    assert tuple(got.shape) == (n, cout, h, w) and got.dtype == torch.float32 and got.is_contiguous(memory_format=CL)
    assert rel_err(got.cpu(), want.cpu()) <= 2e-6
    if copy16:
        assert tuple(got16.shape) == (n, 8, h, w) and got16.dtype == tdt
        assert torch.equal(got16[:, :cout].float(), got.to(tdt).float()) and not got16[:, cout:].any()
        _, got4 = ops.mconv2d_pair_thin(x, mwa, mwb, pad_mode=pad_mode, act_b=act, copy16=True, copy16_channels=4)
        assert tuple(got4.shape) == (n, 4, h, w) and torch.equal(got4, got16[:, :4])
    # against torch on the same 16-bit-rounded intermediate
    pm = "replicate" if pad_mode == "replicate" else "constant"
    cv = lambda t, w_, b_: F.conv2d(F.pad(t, (1, 1, 1, 1), mode=pm), w_, b_)
    ref = cv(F.relu(cv(x.float().cpu(), wa, mwa.b.cpu())).to(tdt).float(), wb, mwb.b.cpu())
    ref = {"sigmoid": torch.sigmoid, "tanh": torch.tanh, "relu": F.relu, "none": lambda t: t}[act](ref)
    assert rel_err(got.cpu(), ref) <= (3e-3 if dt == "f16" else 2e-2)
    assert not ops.mconv2d_pair_thin_applicable(x, mwa, ops.MConvWeights(torch.randn(24, 64, 3, 3, generator=g), None, [64], tdt, "cuda"))


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("pad_mode", ["zeros", "replicate"])
@pytest.mark.parametrize("shape", [(2, 38, 150), (1, 201, 64), (3, 7, 131), (2, 2, 66), (1, 3, 254), (1, 132, 126), (1, 133, 127), (2, 270, 480), (1, 540, 960)],
                         ids=lambda s: "x".join(map(str, s)))
def test_stacked_conv_32_64_64_with_pooled_output(shape, pad_mode, dt):
    """ie_mconv_pair_pool_nhwc (U-Net-32's `down1`: DoubleConv(32, 64) and the MaxPool2d(2) of `down2`, unet.py:43-48,70-72) in one launch: the skip tensor
    and its pooled copy carry the same bits as ie_mconv_nhwc + ie_mconv_nhwc(pool) (odd sizes: the last row / column is not pooled, as nn.MaxPool2d(2))."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(43)
    x = torch.randn(n, 32, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL)
    wa = (torch.randn(64, 32, 3, 3, generator=g) / 17.0).to(tdt).float()
    wb = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(tdt).float()
    mwa = ops.MConvWeights(wa, torch.randn(64, generator=g) * 0.2, [32], tdt, "cuda")
    mwb = ops.MConvWeights(wb, torch.randn(64, generator=g) * 0.2, [64], tdt, "cuda")
    assert ops.mconv2d_pair_pool_applicable(x, mwa, mwb, pad_mode=pad_mode)
    y, yp = ops.mconv2d_pair_pool(x, mwa, mwb, pad_mode=pad_mode)
    mid = ops.mconv2d(x, mwa, pad_mode=pad_mode, act="relu")
    want, want_p = ops.mconv2d(mid, mwb, pad_mode=pad_mode, act="relu", pool=True)
    assert tuple(y.shape) == (n, 64, h, w) and tuple(yp.shape) == (n, 64, h // 2, w // 2) and y.dtype == tdt and yp.is_contiguous(memory_format=CL)
    assert torch.equal(y, want)
    assert torch.equal(yp, want_p)
    assert torch.equal(yp.float(), F.max_pool2d(y.float(), 2))
    pm = "replicate" if pad_mode == "replicate" else "constant"
    cv = lambda t, w_, b_: F.conv2d(F.pad(t, (1, 1, 1, 1), mode=pm), w_, b_)
    ref = F.relu(cv(F.relu(cv(x.float().cpu(), wa, mwa.b.cpu())).to(tdt).float(), wb, mwb.b.cpu()))
    assert rel_err(y.float().cpu(), ref) <= (3e-3 if dt == "f16" else 2e-2)
    assert not ops.mconv2d_pair_pool_applicable(torch.zeros(1, 64, 8, 64, dtype=tdt, device="cuda").contiguous(memory_format=CL), mwb, mwb)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("classes,clamp", [(3, (0.0, 1.0)), (16, None), (1, (-0.5, 0.5))])
@pytest.mark.parametrize("shape", [(2, 38, 150), (1, 200, 64), (3, 6, 130), (2, 2, 66), (1, 2, 254), (1, 132, 126), (1, 270, 480)], ids=lambda s: "x".join(map(str, s)))
def test_unet_last_level_in_one_launch(shape, classes, clamp, dt):
    """ie_unet_up_head_nhwc (unet.py:75-112 `Up.forward` after the up-sampling, :186-195 `outc`): cat([skip, up]) -> DoubleConv(64, 32, mid 32) -> Conv1x1 ->
    clamp -> NCHW fp32 in one launch equals the separate launches (ie_mconv_nhwc + ie_mconv_tail_nhwc): the same 16-bit roundings of both
    intermediates, fp32 accumulation in another order."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(41)
    skip = torch.randn(n, 32, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL)
    up = torch.randn(n, 32, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL)
    wa = (torch.randn(32, 64, 3, 3, generator=g) / 24.0).to(tdt).float()
    wb = (torch.randn(32, 32, 3, 3, generator=g) / 17.0).to(tdt).float()
    mwa = ops.MConvWeights(wa, torch.randn(32, generator=g) * 0.2, [32, 32], tdt, "cuda")
    mwb = ops.MConvWeights(wb, torch.randn(32, generator=g) * 0.2, [32], tdt, "cuda")
    w1 = (torch.randn(classes, 32, 1, 1, generator=g) / 32 ** 0.5).to(tdt).float()
    b1 = torch.randn(classes, generator=g) * 0.1
    tail = ops.MTailWeights(w1, b1, tdt, "cuda")
    assert ops.unet_up_head_applicable(skip, up, mwa, mwb, tail)
    got = ops.unet_up_head(skip, up, mwa, mwb, tail, clamp=clamp)
    assert tuple(got.shape) == (n, classes, h, w) and got.dtype == torch.float32 and got.is_contiguous()
    mid = ops.mconv2d([skip, up], mwa, act="relu")
    x2 = ops.mconv2d(mid, mwb, act="relu")
    want = ops.to_nchw(ops.mconv2d(x2, ops.MConvWeights(w1, b1, [32], tdt, "cuda"), pad=0, clamp=clamp, out_f32=True))
    assert rel_err(got.cpu(), want.cpu()) <= 2e-6
    ref = F.relu(F.conv2d(torch.cat([skip.float().cpu(), up.float().cpu()], 1), wa, mwa.b.cpu(), padding=1)).to(tdt).float()
    ref = F.conv2d(F.relu(F.conv2d(ref, wb, mwb.b.cpu(), padding=1)).to(tdt).float(), w1, b1)
    ref = ref if clamp is None else ref.clamp(*clamp)
    assert rel_err(got.cpu(), ref) <= (3e-3 if dt == "f16" else 2e-2)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 38, 150), (1, 200, 64), (3, 6, 130), (2, 4, 66), (1, 4, 254), (1, 132, 126), (1, 270, 480), (1, 64, 504)], ids=lambda s: "x".join(map(str, s)))
def test_unet_last_level_with_the_upsampling_inside(shape, dt):
    """ie_unet_up_head_nhwc with a HALF-resolution second source (unet.py:82-83, 97-112: nn.Upsample(scale_factor=2, bilinear, align_corners=True), cat,
    DoubleConv, outc): the bilinear x2 up-sampling happens while the kernel fills its LDS rows.  Equals the up-sampling launch + the fused level up to
    the rare 16-bit rounding flip of an up-sampled value (fp32 weights in another association), and torch on the same 16-bit inputs."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(43)
    skip = torch.randn(n, 32, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL)
    low = torch.randn(n, 32, h // 2, w // 2, generator=g).to(tdt).cuda().contiguous(memory_format=CL)
    wa = (torch.randn(32, 64, 3, 3, generator=g) / 24.0).to(tdt).float()
    wb = (torch.randn(32, 32, 3, 3, generator=g) / 17.0).to(tdt).float()
    mwa = ops.MConvWeights(wa, torch.randn(32, generator=g) * 0.2, [32, 32], tdt, "cuda")
    mwb = ops.MConvWeights(wb, torch.randn(32, generator=g) * 0.2, [32], tdt, "cuda")
    w1 = (torch.randn(3, 32, 1, 1, generator=g) / 32 ** 0.5).to(tdt).float()
    b1 = torch.randn(3, generator=g) * 0.1
    tail = ops.MTailWeights(w1, b1, tdt, "cuda")
    assert ops.unet_up_head_applicable(skip, low, mwa, mwb, tail)
    got = ops.unet_up_head(skip, low, mwa, mwb, tail, clamp=(0.0, 1.0))
    up = ops.upsample2x_bilinear(low)
    want = ops.unet_up_head(skip, up, mwa, mwb, tail, clamp=(0.0, 1.0))
    assert tuple(got.shape) == (n, 3, h, w)
    # fp16: the in-kernel blend takes its four weights as fp16 (v_dot2_f32_f16): an up-sampled value differs from ie_resize_nhwc's by up to one fp16 ulp;
    # bf16: fp32 weights, a flipped up-sampled value is 4e-3 of itself
    assert rel_err(got.cpu(), want.cpu()) <= (2e-3 if dt == "f16" else 2e-2)
    upr = F.interpolate(low.float().cpu(), scale_factor=2, mode="bilinear", align_corners=True).to(tdt).float()
    ref = F.relu(F.conv2d(torch.cat([skip.float().cpu(), upr], 1), wa, mwa.b.cpu(), padding=1)).to(tdt).float()
    ref = F.conv2d(F.relu(F.conv2d(ref, wb, mwb.b.cpu(), padding=1)).to(tdt).float(), w1, b1).clamp(0.0, 1.0)
    assert rel_err(got.cpu(), ref) <= (3e-3 if dt == "f16" else 2e-2)
    odd = torch.randn(n, 32, h // 2 - 1, w // 2, generator=g).to(tdt).cuda().contiguous(memory_format=CL)
    assert not ops.unet_up_head_applicable(skip, odd, mwa, mwb, tail)       # not exactly half: the caller up-samples and pads (unet.py:99-104)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 37, 150), (1, 200, 64), (3, 6, 129), (2, 1, 66), (1, 3, 253), (1, 131, 30), (1, 270, 480)], ids=lambda s: "x".join(map(str, s)))
def test_first_9x9_conv_on_a_4_channel_image(shape, dt):
    """ie_conv9x9_c4_nhwc (DecomNet's `net1_conv0`, retinexnet/model.py:26: Conv2d(4, 64, 9, padding=4, padding_mode="replicate")): the strip kernel with nine
    rotating accumulator sets against torch on the same 16-bit-rounded inputs and weights, and against the tiled kernel on the 8-channel carrier."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(47)
    img = torch.rand(n, 3, h, w, generator=g)
    x4 = ops.retinex_decom_input(img.cuda(), tdt, 4)
    x8 = ops.retinex_decom_input(img.cuda(), tdt, 8)
    assert tuple(x4.shape) == (n, 4, h, w) and torch.equal(x4, x8[:, :4])
    wt = (torch.randn(64, 4, 9, 9, generator=g) / 18.0).to(tdt).float()
    bias = torch.randn(64, generator=g) * 0.2
    got = ops.conv9x9_c4(x4, ops.Conv9x9C4Weights(wt, bias, tdt, "cuda"))
    assert tuple(got.shape) == (n, 64, h, w) and got.dtype == tdt and got.is_contiguous(memory_format=CL)
    ref = F.conv2d(F.pad(x4.float().cpu(), (4, 4, 4, 4), mode="replicate"), wt, bias)
    assert rel_err(got.float().cpu(), ref) <= (1e-3 if dt == "f16" else 8e-3)          # one rounding of the stored result
    mw = ops.MConvWeights(wt, bias, [8], tdt, "cuda", cin_map=[0, 1, 2, 3, -1, -1, -1, -1])
    old = ops.mconv2d(x8, mw, pad=4, pad_mode="replicate")
    assert rel_err(got.float().cpu(), old.float().cpu()) <= (1e-3 if dt == "f16" else 8e-3)
    # the same kernel at 3x3 (RelightNet's first layer, model.py:58)
    w3 = (torch.randn(64, 4, 3, 3, generator=g) / 6.0).to(tdt).float()
    got3 = ops.conv9x9_c4(x4, ops.Conv9x9C4Weights(w3, bias, tdt, "cuda"))
    ref3 = F.conv2d(F.pad(x4.float().cpu(), (1, 1, 1, 1), mode="replicate"), w3, bias)
    assert rel_err(got3.float().cpu(), ref3) <= (1e-3 if dt == "f16" else 8e-3)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("pad_mode", ["zeros", "replicate"])
@pytest.mark.parametrize("cout,act,scales", [(1, "none", (4, 2, 1)), (4, "sigmoid", (1,)), (3, "tanh", (2, 1)), (2, "relu", (1, 1, 1))])
@pytest.mark.parametrize("shape", [(2, 36, 152), (1, 200, 64), (3, 8, 132), (2, 4, 68), (1, 4, 256), (1, 132, 128)], ids=lambda s: "x".join(map(str, s)))
def test_thin_output_conv_over_sources_at_their_own_resolution(shape, cout, act, scales, pad_mode, dt):
    """ie_mconv_nhwc's thin-output strip form (ie_conv_thin3.hip; RelightNet's tail, retinexnet/model.py:83-88: F.interpolate(nearest) of the coarser decoder
    levels, cat, 1x1 fusion + 3x3 output composed into one 3x3 conv 192 -> 1): up to three 64-channel sources, each at 1 / scale of the output size, 1 .. 4
    output channels in fp32 -- against torch on the same 16-bit inputs and weights."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(53)
    srcs = [torch.randn(n, 64, h // sc, w // sc, generator=g).to(tdt).cuda().contiguous(memory_format=CL) for sc in scales]
    cin = 64 * len(scales)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(tdt).float()
    bias = torch.randn(cout, generator=g) * 0.3
    mw = ops.MConvWeights(wt, bias, [64] * len(scales), tdt, "cuda")
    got = ops.mconv2d(srcs, mw, pad_mode=pad_mode, act=act, out_f32=True)
    assert tuple(got.shape) == (n, cout, h, w) and got.dtype == torch.float32
    up = torch.cat([F.interpolate(t.float().cpu(), size=(h, w), mode="nearest") for t in srcs], 1)
    ref = F.conv2d(F.pad(up, (1, 1, 1, 1), mode="replicate" if pad_mode == "replicate" else "constant"), wt, bias)
    ref = {"sigmoid": torch.sigmoid, "tanh": torch.tanh, "relu": F.relu, "none": lambda t: t}[act](ref)
    assert rel_err(got.cpu(), ref) <= 2e-5                                    # fp32 accumulation of exact 16-bit products, fp32 out


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 38, 150), (1, 200, 64), (3, 6, 130), (2, 2, 66), (1, 3, 254), (1, 132, 126), (1, 270, 480), (1, 64, 500)], ids=lambda s: "x".join(map(str, s)))
def test_unet_first_level_from_the_nchw_image(shape, dt):
    """ie_unet_first_pool_nchw (unet.py:43-48 `inc`, :70-72 the first MaxPool2d(2)): conv1 + conv2 + pool from the caller's fp32 NCHW image in one launch equals the
    layout pass + ie_mconv_first_pool_nhwc to fp32 rounding of conv1 (another K order), the pooled copy is F.max_pool2d of ITS full-resolution output bit for bit,
    and both agree with torch on the same 16-bit operands."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(59)
    img = torch.rand(n, 3, h, w, generator=g)
    w1 = (torch.randn(32, 3, 3, 3, generator=g) / 27 ** 0.5).to(tdt).float()
    w2 = (torch.randn(32, 32, 3, 3, generator=g) / 17.0).to(tdt).float()
    b1, b2 = torch.randn(32, generator=g) * 0.2, torch.randn(32, generator=g) * 0.2
    fw = ops.UNetFirstWeights(w1, b1, tdt, "cuda")
    mw2 = ops.MConvWeights(w2, b2, [32], tdt, "cuda")
    assert ops.unet_first_pool_applicable(img.cuda(), fw, mw2)
    out, pooled = ops.unet_first_pool(img.cuda(), fw, mw2)
    assert tuple(out.shape) == (n, 32, h, w) and tuple(pooled.shape) == (n, 32, h // 2, w // 2) and out.dtype == tdt
    assert torch.equal(pooled.float().cpu(), F.max_pool2d(out.float().cpu(), 2))
    x8 = ops.to_nhwc16(img.cuda(), tdt, 8)
    mw1 = ops.MConvWeights(w1, b1, [8], tdt, "cuda", cin_map=[0, 1, 2, -1, -1, -1, -1, -1])
    want, want_p = ops.mconv2d_first_pool(x8, mw1, mw2)
    tol = 2e-3 if dt == "f16" else 1.6e-2          # a 1-ulp difference of the 16-bit intermediate (conv1 summed in another order) through conv2
    assert rel_err(out.float().cpu(), want.float().cpu()) <= tol and rel_err(pooled.float().cpu(), want_p.float().cpu()) <= tol
    ref = F.relu(F.conv2d(F.relu(F.conv2d(x8[:, :3].float().cpu(), w1, b1, padding=1)).to(tdt).float(), w2, b2, padding=1))
    assert rel_err(out.float().cpu(), ref) <= (3e-3 if dt == "f16" else 2e-2)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 37, 150), (1, 200, 64), (3, 6, 129), (1, 270, 480), (2, 2, 66)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("c", [32, 64])
def test_first_double_conv_with_pool(c, dt, shape):
    """ie_mconv_first_pool_nhwc (unet.py:43-48, 70-72, 152): first conv (8-channel image -> C) + second conv (C -> C) + 2x2 max-pool in one
    launch equals the thin-kernel conv followed by the strip conv with its fused pool -- the same contraction order, roundings and padding
    (the intermediate's rows outside the image are zeros, not conv1 of the padding), so bit for bit."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(31)
    img = torch.rand(n, 3, h, w, generator=g)
    x8 = ops.to_nhwc16(img.cuda(), tdt, 8)
    w1 = (torch.randn(c, 3, 3, 3, generator=g) / 27 ** 0.5).to(tdt).float()
    w2 = (torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5).to(tdt).float()
    mw1 = ops.MConvWeights(w1, torch.randn(c, generator=g) * 0.2, [8], tdt, "cuda", cin_map=[0, 1, 2, -1, -1, -1, -1, -1])
    mw2 = ops.MConvWeights(w2, torch.randn(c, generator=g) * 0.2, [c], tdt, "cuda")
    assert ops.mconv2d_first_pool_applicable(x8, mw1, mw2)
    out, pooled = ops.mconv2d_first_pool(x8, mw1, mw2)
    mid = ops.mconv2d(x8, mw1, act="relu")
    want, want_p = ops.mconv2d(mid, mw2, act="relu", pool=True)
    assert torch.equal(out, want) and torch.equal(pooled, want_p)
    ref = F.relu(F.conv2d(F.relu(F.conv2d(x8[:, :3].float().cpu(), w1, mw1.b.cpu(), padding=1)).to(tdt).float(), w2, mw2.b.cpu(), padding=1))
    assert rel_err(out.float().cpu(), ref) <= (3e-3 if dt == "f16" else 2e-2)


def test_fused_1x1_tail_refusals():
    g = torch.Generator().manual_seed(1)
    mk = lambda c, w: torch.randn(1, c, 8, w, generator=g).half().cuda().contiguous(memory_format=CL)
    mw = ops.MConvWeights(torch.randn(32, 32, 3, 3, generator=g), None, [32], torch.float16, "cuda")
    tail = ops.MTailWeights(torch.randn(3, 32, 1, 1, generator=g), None, torch.float16, "cuda")
    assert not ops.mconv2d_tail_applicable(mk(32, 150), mw, tail)                 # W % 4 != 0
    assert not ops.mconv2d_tail_applicable(mk(32, 152), mw, tail, act="none")     # the tail contracts ReLU'd rows
    wide = ops.MConvWeights(torch.randn(128, 32, 3, 3, generator=g), None, [32], torch.float16, "cuda")
    assert not ops.mconv2d_tail_applicable(mk(32, 152), wide, ops.MTailWeights(torch.randn(3, 128, 1, 1, generator=g), None, torch.float16, "cuda"))
    with pytest.raises(capi.IEError):
        ops.mconv2d_tail(mk(32, 150), mw, tail)
    with pytest.raises(capi.IEError):
        ops.MTailWeights(torch.randn(17, 32, 1, 1), None, torch.float16, "cuda")


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("c", CASES, ids=_id)
def test_mconv_vs_torch(c, dt, shape=(2, 37, 45)):
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape                                                    # default: ragged tiles in both directions
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(n, ci, h, w, generator=g).to(tdt) for ci in c["cs"]]
    cin, k, s, p = sum(c["cs"]), c["k"], c.get("s", 1), c["p"]
    wt = (torch.randn(c["co"], cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(tdt).float()
    b = None if c.get("bias") is False else torch.randn(c["co"], generator=g)
    mode = c.get("mode", "zeros")
    xcat = torch.cat([x.float() for x in xs], 1)
    xp = F.pad(xcat, (p,) * 4, mode="replicate") if mode == "replicate" and p > 0 else xcat
    ref = F.conv2d(xp, wt, b, stride=s, padding=0 if mode == "replicate" else p)
    act = c.get("act", "none")
    ref = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2), "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act](ref)
    res = None
    if c.get("residual"):
        res = torch.randn(ref.shape, generator=g).to(tdt)
        ref = ref + res.float()
    if c.get("clamp"):
        ref = ref.clamp(*c["clamp"])
    mw = ops.MConvWeights(wt, b, c["cs"], tdt, "cuda")
    out = ops.mconv2d([x.cuda().contiguous(memory_format=CL) for x in xs], mw, stride=s, pad=p, pad_mode=mode, act=act, slope=0.2,
                      residual=None if res is None else res.cuda().contiguous(memory_format=CL), clamp=c.get("clamp"),
                      out_f32=c.get("out_f32", False))
    assert tuple(out.shape) == tuple(ref.shape)
    assert out.dtype == (torch.float32 if c.get("out_f32") else tdt)
    tol = 2e-5 if c.get("out_f32") else (1.5e-3 if dt == "f16" else 1e-2)     # stored-result rounding: 2^-11 / 2^-8 relative
    assert rel_err(out.float().cpu(), ref) <= tol


def test_mconv_input_channel_map_and_errors():
    g = torch.Generator().manual_seed(3)
    w = torch.randn(16, 3, 3, 3, generator=g).half().float()
    x = torch.rand(1, 3, 20, 24, generator=g)
    x16 = ops.to_nhwc16(x.cuda(), torch.float16)                       # RGB -> 8 channels, zeros appended
    assert x16.shape[1] == 8 and float(x16[:, 3:].abs().max()) == 0.0
    mw = ops.MConvWeights(w, None, [8], torch.float16, "cuda", cin_map=[0, 1, 2, -1, -1, -1, -1, -1])
    out = ops.mconv2d(x16, mw)
    ref = F.conv2d(x.half().float(), w, None, padding=1)
    assert rel_err(out.float().cpu(), ref) <= 1.5e-3
    with pytest.raises(capi.IEError):
        ops.MConvWeights(w, None, [3], torch.float16, "cuda")          # channel counts must be multiples of 8
    with pytest.raises(capi.IEError):
        ops.mconv2d(x16.to(torch.bfloat16), mw)
    with pytest.raises(capi.IEError):
        ops.mconv2d([x16, x16], mw)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("cs,co,act", [([32], 32, "relu"), ([32, 32], 32, "relu"), ([40, 24], 3, "tanh"), ([16], 72, "none")])
def test_fused_depthwise_pointwise(cs, co, act, dt):
    """ie_dsconv_nhwc (dsconv.py:106-149): depthwise 3x3 of the virtual concat, result rounded to the 16-bit type in LDS,
    pointwise 1x1 on the matrix cores.  Reference: the same two steps in torch with the intermediate rounded the same way."""
    tdt = ops.TORCH_DT[dt]
    g = torch.Generator().manual_seed(5)
    n, h, w = 2, 37, 45
    xs = [torch.randn(n, c, h, w, generator=g).to(tdt) for c in cs]
    cin = sum(cs)
    wd = torch.randn(cin, 1, 3, 3, generator=g) * 0.3
    bd = torch.randn(cin, generator=g) * 0.1
    wp = (torch.randn(co, cin, 1, 1, generator=g) / cin ** 0.5).to(tdt).float()
    bp = torch.randn(co, generator=g)
    mid = F.conv2d(torch.cat([x.float() for x in xs], 1), wd, bd, padding=1, groups=cin).to(tdt).float()
    ref = F.conv2d(mid, wp, bp)
    ref = {"relu": F.relu, "tanh": torch.tanh, "none": lambda t: t}[act](ref)
    pw = ops.MConvWeights(wp, bp, cs, tdt, "cuda")
    out = ops.dsconv2d([x.cuda().contiguous(memory_format=CL) for x in xs], wd[:, 0].reshape(cin, 9).contiguous().cuda(), bd.cuda(), pw, act=act,
                       out_f32=(co == 3))
    tol = 3e-3 if dt == "f16" else 2e-2          # a 1-ulp difference in the rounded intermediate is amplified by the 1x1 contraction
    assert tuple(out.shape) == tuple(ref.shape) and rel_err(out.float().cpu(), ref) <= tol
    with pytest.raises(capi.IEError):
        ops.dsconv2d([xs[0].cuda().contiguous(memory_format=CL)[:, :8].contiguous(memory_format=CL)], wd[:8, 0].reshape(8, 9).cuda(), None,
                     ops.MConvWeights(wp[:, :8].contiguous(), bp, [8], tdt, "cuda"))


@pytest.mark.parametrize("cs,co,k", [([64], 64, 3), ([32, 40], 72, 3), ([128], 24, 1), ([16, 16, 16], 16, 3)])
def test_device_weight_packing_matches_host(cs, co, k):
    """ie_mconv_pack_weights_dev (training: master weights change every step) produces the bytes of the host packer, for the forward
    conv and for the data-gradient conv of every source (flipped, transposed weights)."""
    g = torch.Generator().manual_seed(9)
    cin = sum(cs)
    w = torch.randn(co, cin, k, k, generator=g)
    hwio = ops.ConvWeights(w, None, "cuda").w                       # [k][k][cin][co] fp32 on the device
    for tdt in (torch.float16, torch.bfloat16):
        host = ops.MConvWeights(w, None, cs, tdt, "cuda")
        dev = ops.MConvWeightsDev(hwio, None, cs, co, k, k, tdt)
        assert torch.equal(host.w, dev.w)
        base = 0
        for si, c in enumerate(cs):
            wt = w[:, base:base + c].flip(2, 3).permute(1, 0, 2, 3).contiguous()     # OIHW of the gradient conv: (c, co, k, k)
            host_g = ops.MConvWeights(wt, None, [co], tdt, "cuda") if co % 8 == 0 else None
            if host_g is not None:
                dev_g = ops.MConvWeightsDev(hwio, None, cs, co, k, k, tdt, mode="dgrad", src_index=si)
                assert dev_g.cout == c and dev_g.split == [co] and torch.equal(host_g.w, dev_g.w)
            base += c


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("ci,c,shape", [(64, 32, (2, 37, 45)), (128, 64, (1, 20, 70)), (32, 16, (2, 9, 33))])
def test_fused_pixel_shuffle(ci, c, shape, dt):
    """Upsampler of the PixelShuffle decoder (edsr.py:88-123): conv3x3(ci -> 4c) + nn.PixelShuffle(2) in one kernel -- the weight rows are
    packed in [i][j][c] order and the epilogue scatters every 16-channel fragment to its sub-pixel."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn(n, ci, h, w, generator=g).to(tdt)
    wt = (torch.randn(4 * c, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).to(tdt).float()
    b = torch.randn(4 * c, generator=g)
    ref = F.pixel_shuffle(F.relu(F.conv2d(x.float(), wt, b, padding=1)), 2)
    mw = ops.MConvWeights(wt, b, [ci], tdt, "cuda", shuffle=2)
    out = ops.mconv2d(x.cuda().contiguous(memory_format=CL), mw, act="relu")
    assert tuple(out.shape) == (n, c, 2 * h, 2 * w) and out.dtype == tdt
    assert rel_err(out.float().cpu(), ref) <= (1.5e-3 if dt == "f16" else 1e-2)
    if (4 * c) % 64:
        return
    with pytest.raises(capi.IEError):                                   # 1x1 convs do not take the fused form
        ops.mconv2d(x.cuda().contiguous(memory_format=CL), ops.MConvWeights(wt[:, :, :1, :1].contiguous(), b, [ci], tdt, "cuda", shuffle=2), pad=0)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("case", [
    dict(cs=[64, 64], co=64, k=3, p=1, mode="replicate", act="relu", hw=(36, 150), small=[(18, 75), None]),     # retinexnet/model.py:79-80 (strip kernel)
    dict(cs=[32, 32], co=32, k=3, p=1, mode="zeros", act="none", hw=(21, 131), small=[None, (7, 44)]),           # odd ratios
    dict(cs=[64, 64], co=64, k=3, p=1, mode="replicate", act="relu", hw=(20, 40), small=[(10, 20), None]),       # narrow image: tiled kernel
    dict(cs=[64, 64, 64], co=64, k=1, p=1, mode="replicate", act="none", hw=(24, 40), small=[(6, 10), (12, 20), None]),   # model.py:84-87 fusion
], ids=["strip-c128", "strip-c64-odd", "tiled-3x3", "fusion-1x1"])
def test_nearest_upsampled_sources(case, dt):
    """A source smaller than the conv's image is read through F.interpolate(x, size=...) (nearest) folded into the kernel's addressing."""
    tdt = ops.TORCH_DT[dt]
    n, (h, w) = 2, case["hw"]
    g = torch.Generator().manual_seed(23)
    xs = [torch.randn(n, c, *(sm if sm else (h, w)), generator=g).to(tdt) for c, sm in zip(case["cs"], case["small"])]
    cin, k, p = sum(case["cs"]), case["k"], case["p"]
    wt = (torch.randn(case["co"], cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(tdt).float()
    b = torch.randn(case["co"], generator=g)
    xcat = torch.cat([F.interpolate(x.float(), size=(h, w)) if sm else x.float() for x, sm in zip(xs, case["small"])], 1)
    xp = F.pad(xcat, (p,) * 4, mode="replicate") if case["mode"] == "replicate" else xcat
    ref = F.conv2d(xp, wt, b, padding=0 if case["mode"] == "replicate" else p)
    ref = F.relu(ref) if case["act"] == "relu" else ref
    mw = ops.MConvWeights(wt, b, case["cs"], tdt, "cuda")
    out = ops.mconv2d([x.cuda().contiguous(memory_format=CL) for x in xs], mw, pad=p, pad_mode=case["mode"], act=case["act"])
    assert tuple(out.shape) == tuple(ref.shape)
    assert rel_err(out.float().cpu(), ref) <= (1.5e-3 if dt == "f16" else 1e-2)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("act", ["none", "relu", "lrelu"])
@pytest.mark.parametrize("cs,co", [([64], 64), ([192], 64), ([24, 40], 48), ([128, 64, 16], 272), ([16], 16), ([72], 32), ([512], 128)], ids=lambda v: str(v).replace(" ", ""))
@pytest.mark.parametrize("shape", [(2, 37, 150), (1, 5, 7), (1, 128, 129), (3, 16, 8)], ids=lambda s: "x".join(map(str, s)))
def test_1x1_conv_as_streaming_gemm(shape, cs, co, act, dt):
    """1x1 layers (HINet's identity / csff convs hinet.py:60-83, LLUnet++'s skip convs llunetpp.py:55-70, RelightNet's fusion conv retinexnet/model.py:66) run on
    ie_conv1x1.hip: 128-pixel rows of the [pixel][Cin] matrix through an LDS double buffer.  Against torch fp32 on the same 16-bit operands; ragged last
    workgroups, short last K chunks (24, 40, 16, 72 channels), several cout tiles, all three wave layouts."""
    tdt = ops.TORCH_DT[dt]
    n, h, w = shape
    g = torch.Generator().manual_seed(53)
    xs = [torch.randn(n, c, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL) for c in cs]
    cin = sum(cs)
    wt = (torch.randn(co, cin, 1, 1, generator=g) / cin ** 0.5).to(tdt).float()
    b = torch.randn(co, generator=g) * 0.3
    mw = ops.MConvWeights(wt, b, cs, tdt, "cuda")
    kw = dict(act=act, slope=0.2) if act == "lrelu" else dict(act=act)
    y = ops.mconv2d(xs, mw, pad=0, **kw)
    assert tuple(y.shape) == (n, co, h, w) and y.dtype == tdt and y.is_contiguous(memory_format=CL)
    ref = F.conv2d(torch.cat([x.float().cpu() for x in xs], 1), wt, b)
    ref = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2)}[act](ref)
    assert rel_err(y.float().cpu(), ref) <= (2e-3 if dt == "f16" else 1.2e-2)

