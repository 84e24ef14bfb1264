"""GPU parity of the generator models (fp32 NHWC building blocks through the C ABI) against goldens and
the CPU oracle.  Tolerance 1e-4 on max|a-b|/max|b| (bound asked: 1e-3)."""
import ast
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mon_amd
from mon_amd import ops
from oracle import generators as og
from tests.helpers import GOLDEN, rel_err, seeded_bn_state_dict, seeded_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "zero_dcepp_*.npz"))), ids=lambda p: os.path.basename(p)[11:-4])
def test_zero_dcepp_golden(path):
    g = np.load(path)
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    m = mon_amd.MODELS.build(name="zero_dce++_re", scale_factor=float(g["scale_factor"]), weights=sd).to("cuda")
    out = m.forward({"image": torch.from_numpy(g["image"]).cuda()})
    assert list(out.keys()) == ["adjust", "enhanced"]
    assert rel_err(out["enhanced"].cpu(), g["enhanced"]) <= TOL
    assert rel_err(out["adjust"].cpu(), g["adjust"]) <= TOL


def test_llunetpp_golden_and_infer():
    g = np.load(os.path.join(GOLDEN, "llunetpp_seed31_1x32x48.npz"))
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["key_order"], g["key_shapes"])}
    sd = seeded_state_dict(shapes, int(g["seed"]))
    m = mon_amd.MODELS.build(name="llunet++_re", weights=sd).to("cuda")
    x = torch.from_numpy(g["image"])
    out = m.forward({"image": x.cuda()})["enhanced"].cpu()
    assert rel_err(out, g["enhanced"]) <= TOL
    with pytest.raises(ValueError):                         # 1080-like sizes only work through infer
        m.forward({"image": torch.rand(1, 3, 40, 48).cuda()})
    y = torch.rand(1, 3, 40, 56, generator=torch.Generator().manual_seed(1))
    got = m.infer({"image": y.clone()})["enhanced"].cpu()
    yr = F.interpolate(y, size=(64, 64), mode="bilinear", align_corners=False)
    ref = F.interpolate(og.llunetpp_forward(m.state_dict(), yr)["enhanced"], size=(40, 56), mode="bilinear", align_corners=False)
    assert tuple(got.shape) == (1, 3, 40, 56) and rel_err(got, ref) <= TOL


@pytest.mark.parametrize("cfg", [dict(base_channels=64, num_classes=34, enhance=False, upsampler="bilinear", hw=(33, 47)),
                                 dict(base_channels=32, num_classes=3, enhance=True, upsampler="bilinear", hw=(135, 70)),
                                 dict(base_channels=32, num_classes=3, enhance=True, upsampler="pixelshuffle", hw=(48, 64))],
                         ids=["unet64-seg-odd", "unet32-enh-odd", "unet32-pixelshuffle"])
def test_unet_vs_oracle(cfg):
    hw = cfg.pop("hw")
    m = mon_amd.MODELS.build(name="unet", seed=3, **cfg).to("cuda")
    x = torch.rand(2, 3, *hw, generator=torch.Generator().manual_seed(2))
    out = m.forward({"image": x.cuda()})
    key = "enhanced" if cfg["enhance"] else "semantic"
    ref = og.unet_forward(m.state_dict(), x, cfg["upsampler"], cfg["enhance"])[key]
    assert tuple(out[key].shape) == tuple(ref.shape)
    assert rel_err(out[key].cpu(), ref) <= TOL


def _unet_golden(path):
    g = np.load(path)
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["key_order"], g["key_shapes"])}
    return g, seeded_bn_state_dict(shapes, int(g["seed"])), "bilinear" if bool(g["bilinear"]) else "convtranspose"


@pytest.mark.parametrize("dt,tol", [("f32", TOL), ("f16", 1e-3), ("bf16", 1.5e-2)])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "unet_mtfe_*.npz"))), ids=lambda p: os.path.basename(p)[10:-4])
def test_unet_golden(path, dt, tol):
    """The engine's U-Net against the outputs of the reference tree's own runnable U-Net (mtfe/unet_model.py; same topology as
    unet.py:35-197), checkpoint in the original key spelling, both decoders, odd frames."""
    g, sd_ref, upsampler = _unet_golden(path)
    x = torch.from_numpy(g["image"]).cuda()
    m = mon_amd.MODELS.build(name="unet", base_channels=16, num_classes=3, conv_bias=True, upsampler=upsampler, weights=dict(sd_ref),
                             compute_dtype=dt).to("cuda")
    out = m.forward({"image": x})["semantic"].cpu()
    assert tuple(out.shape) == g["logits"].shape and rel_err(out, g["logits"]) <= tol
    e = mon_amd.MODELS.build(name="unet", base_channels=16, num_classes=3, conv_bias=True, upsampler=upsampler, weights=dict(sd_ref),
                             compute_dtype=dt, enhance=True).to("cuda")
    enh = e.forward({"image": x})["enhanced"].cpu()
    assert rel_err(enh, np.clip(g["logits"], 0, 1)) <= tol and float(enh.min()) == 0.0


@pytest.mark.parametrize("dt,tol", [("f32", TOL), ("f16", 1e-3), ("bf16", 1.5e-2)])
def test_rrdnet_golden(dt, tol):
    g = np.load(os.path.join(GOLDEN, "rrdnet_seed71_2x37x53.npz"))
    sd_ref = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    m = mon_amd.MODELS.build(name="rrdnet_re", weights=dict(sd_ref), compute_dtype=dt).to("cuda")
    x = torch.from_numpy(g["image"])
    out = m.forward({"image": x.cuda()})
    for k in ("illumination", "reflectance", "noise"):
        assert rel_err(out[k].cpu(), g[k]) <= tol, k
    ill, noise = torch.from_numpy(g["illumination"]), torch.from_numpy(g["noise"])
    want = torch.clamp(torch.pow(ill, 0.4) * ((x - noise) / ill), 0, 1)          # rrdnet.py:271-279 on the reference's own maps
    assert rel_err(out["enhanced"].cpu(), want) <= tol


def test_unet_loads_reference_style_checkpoint(tmp_path):
    """BatchNorm checkpoints carry num_batches_tracked; strict loading ignores exactly those."""
    m = mon_amd.UNet(seed=5, base_channels=32, num_classes=3, enhance=True)
    sd = m.state_dict()
    sd["inc.norm1.num_batches_tracked"] = torch.tensor(7)
    p = tmp_path / "unet.pt"
    torch.save(sd, p)
    m2 = mon_amd.MODELS.build(name="unet", weights=str(p), base_channels=32, num_classes=3, enhance=True).to("cuda")
    x = torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(3))
    assert rel_err(m2.forward({"image": x.cuda()})["enhanced"].cpu(), og.unet_forward(sd, x, "bilinear", True)["enhanced"]) <= TOL


def test_zero_ig_enhance_vs_oracle():
    m = mon_amd.MODELS.build(name="zero_ig_enhance", seed=4, layers=3, channels=64).to("cuda")
    sd = m.state_dict()
    sd = {k: (v * 6 if k.endswith("conv.0.weight") or k.endswith("in_conv.0.weight") else v) for k, v in sd.items()}
    m.load_state_dict(sd)
    x = torch.rand(2, 3, 37, 51, generator=torch.Generator().manual_seed(4))
    out = m.forward({"image": x.cuda()})["enhanced"].cpu()
    ref = og.zero_ig_enhance_forward(sd, x, 3)["enhanced"]
    assert rel_err(out, ref) <= TOL and float(out.min()) >= 0.99e-4 and float(out.max()) <= 1.0


def _zero_ig_golden():
    g = np.load(os.path.join(GOLDEN, "zero_ig_seed51_2x40x56.npz"))
    return g, {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}


@pytest.mark.parametrize("dt,tol", [("f32", TOL), ("f16", 2e-3), ("bf16", 1.5e-2)])
def test_zero_ig_golden(dt, tol):
    """zero_ig_re (Denoise1 -> Enhance -> H2 = image / s2 -> Denoise2) and the Enhance block alone against outputs of the
    reference's own Finetunemodel.forward / Enhancer (tests/golden/make_golden.py); checkpoint in the original key spelling."""
    g, sd = _zero_ig_golden()
    m = mon_amd.MODELS.build(name="zero_ig_re", weights=dict(sd), compute_dtype=dt).to("cuda")
    out = m.forward({"image": torch.from_numpy(g["image"]).cuda()})
    assert list(out.keys()) == ["denoise", "enhanced"]
    assert rel_err(out["enhanced"].cpu(), g["enhanced"]) <= tol
    assert rel_err(out["denoise"].cpu(), g["denoise"]) <= tol
    e = mon_amd.MODELS.build(name="zero_ig_enhance", weights={k[8:]: v for k, v in m.state_dict().items() if k.startswith("enhance.")},
                             compute_dtype=dt).to("cuda")
    assert rel_err(e.forward({"image": torch.from_numpy(g["enhancer_in"]).cuda()})["enhanced"].cpu(), g["enhancer_out"]) <= tol


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "hinet_*.npz"))), ids=lambda p: os.path.basename(p)[6:-4])
def test_hinet_golden(path):
    """hinet_re against both stage outputs of the reference's hinet_arch.HINet (weights regenerated from the seed)."""
    g = np.load(path)
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["key_order"], g["key_shapes"])}
    m = mon_amd.MODELS.build(name="hinet_re", num_channels=int(g["wf"]), weights=seeded_state_dict(shapes, int(g["seed"]))).to("cuda")
    out = m.forward({"image": torch.from_numpy(g["image"]).cuda()})
    assert rel_err(out["stage1"].cpu(), g["stage1"]) <= TOL and rel_err(out["enhanced"].cpu(), g["enhanced"]) <= TOL
    if int(g["wf"]) % 16 == 0:                 # the matrix-core path of the same graph
        for dt, tol in (("f16", 2e-3), ("bf16", 1.5e-2)):
            m16 = mon_amd.MODELS.build(name="hinet_re", num_channels=int(g["wf"]), weights=m.state_dict(), compute_dtype=dt).to("cuda")
            o16 = m16.forward({"image": torch.from_numpy(g["image"]).cuda()})
            assert rel_err(o16["enhanced"].cpu(), g["enhanced"]) <= tol, dt


@pytest.mark.parametrize("dt,tol", [("f32", TOL), ("f16", 2e-3), ("bf16", 1.5e-2)])
def test_rrdnet_and_zsn2n_vs_oracle(dt, tol):
    x = torch.rand(2, 3, 37, 53, generator=torch.Generator().manual_seed(9)) * 0.6 + 0.05
    m = mon_amd.MODELS.build(name="rrdnet_re", seed=11, compute_dtype=dt).to("cuda")
    out, ref = m.forward({"image": x.cuda()}), og.rrdnet_forward(m.state_dict(), x, 0.4)
    assert list(out.keys()) == ["image", "illumination", "reflectance", "noise", "enhanced"]
    for k in ("illumination", "reflectance", "noise", "enhanced"):
        assert tuple(out[k].shape) == tuple(ref[k].shape) and rel_err(out[k].cpu(), ref[k]) <= tol, k
    assert torch.equal(out["image"].cpu(), x)
    for predicting in (True, False):
        z = mon_amd.MODELS.build(name="zsn2n", seed=12, compute_dtype=dt, predicting=predicting).to("cuda")
        sd = {k: v * (3.0 if k.endswith("weight") else 1.0) for k, v in z.state_dict().items()}     # make the clamp bite
        z.load_state_dict(sd)
        got, want = z.forward({"image": x.cuda()})["enhanced"].cpu(), og.zsn2n_forward(sd, x, predicting)["enhanced"]
        assert rel_err(got, want) <= tol, predicting
    assert mon_amd.MODELS.build(name="zsn2n", seed=1).params == 48 * 27 + 48 + 48 * 48 * 9 + 48 + 3 * 48 + 3


def test_new_eltwise_ops_against_torch():
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(2, 3, 9, 11, generator=g) + 0.1, torch.rand(2, 3, 9, 11, generator=g) + 0.2
    one = torch.rand(2, 1, 9, 11, generator=g) + 0.3
    an, bn, on = ops.to_nhwc(a.cuda()), ops.to_nhwc(b.cuda()), ops.to_nhwc(one.cuda())
    assert torch.equal(ops.to_nchw(ops.sub(an, bn)).cpu(), a - b)
    assert torch.equal(ops.to_nchw(ops.div_clamp(an, bn, 0.5, 2.0)).cpu(), (a / b).clamp(0.5, 2.0))
    assert torch.equal(ops.to_nchw(ops.div_clamp(an, on)).cpu(), a / one)
    assert torch.equal(ops.to_nchw(ops.affine(an, 1.0, 1e-4)).cpu(), a + 1e-4)
    assert rel_err(ops.to_nchw(ops.pow_scalar(an, 0.4)).cpu(), a.pow(0.4)) <= 1e-6


@pytest.mark.parametrize("hw", [(40, 56), (45, 67)])
def test_retinexnet_vs_oracle(hw):
    m = mon_amd.MODELS.build(name="retinexnet", seed=6).to("cuda")
    x = torch.rand(1, 3, *hw, generator=torch.Generator().manual_seed(5)) * 0.4
    out = m.forward({"image": x.cuda()})
    ref = og.retinexnet_forward(m.state_dict(), x)
    assert list(out.keys())[-1] == "enhanced"
    for k in ("reflectance", "illumination", "illumination_delta", "enhanced"):
        assert tuple(out[k].shape) == tuple(ref[k].shape), k
        assert rel_err(out[k].cpu(), ref[k]) <= TOL, k


def test_upsampler_reference_golden():
    """The PixelShuffle up-sampler leaf (edsr.py:88-123) against the reference module's own output: fp32 conv + ie_pixel_shuffle_nhwc, and the 16-bit path's ONE kernel
    (conv3x3 with the shuffle in its epilogue, ie_mconv_desc.pixel_shuffle)."""
    g = np.load(os.path.join(GOLDEN, "upsampler_edsr.npz"))
    for tag in ("s2_c16", "s4_c8_relu", "s2_c32_relu"):
        sd = {k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + ".w.")}
        relu = bool(g[f"{tag}.relu"])
        stages = sorted({k.rsplit(".", 1)[0] for k in sd}, key=int)
        a = ops.to_nhwc(torch.from_numpy(g[f"{tag}.x"]).cuda())
        for k in stages:
            cw = ops.ConvWeights(sd[f"{k}.weight"], sd[f"{k}.bias"], "cuda")
            a = ops.pixel_shuffle(ops.conv2d([a], cw, act="relu" if relu else "none"), 2)      # (ReLU commutes with the shuffle)
        assert rel_err(ops.to_nchw(a).cpu(), g[f"{tag}.y"]) <= TOL, tag
        for dt, tol in (("f16", 2e-3), ("bf16", 1.5e-2)):
            tdt = ops.TORCH_DT[dt]
            a = torch.from_numpy(g[f"{tag}.x"]).to(tdt).cuda().contiguous(memory_format=torch.channels_last)
            for k in stages:
                fused = int(sd[f"{k}.weight"].shape[0]) % 64 == 0          # the one-kernel form takes whole 64-channel output tiles; others: conv, then the shuffle pass
                mw = ops.MConvWeights(sd[f"{k}.weight"], sd[f"{k}.bias"], [int(a.shape[1])], tdt, "cuda", shuffle=2 if fused else 0)
                a = ops.mconv2d(a, mw, act="relu" if relu else "none")
                if not fused:
                    a = ops.pixel_shuffle(a, 2)
            assert rel_err(a.float().cpu(), g[f"{tag}.y"]) <= tol, (tag, dt)


def test_learnable_instance_norm_reference_golden():
    """ops.learnable_instance_norm (instance statistics + ONE channel-affine pass, the blend folded into scale and shift) against the reference class's own output
    (tests/golden/learnable_instance_norm.npz, normalization.py:177-224): fp32 and the 16-bit storage types, with and without the fused LeakyReLU LLLINet applies."""
    g = np.load(os.path.join(GOLDEN, "learnable_instance_norm.npz"))
    for tag in ("c12", "c32"):
        t = {k: torch.from_numpy(g[f"{tag}.{k}"]).cuda() for k in ("x", "weight", "bias", "r")}
        y = ops.to_nchw(ops.learnable_instance_norm(ops.to_nhwc(t["x"]), t["weight"], t["bias"], t["r"])).cpu()
        assert rel_err(y, g[f"{tag}.y"]) <= TOL, tag
        ya = ops.to_nchw(ops.learnable_instance_norm(ops.to_nhwc(t["x"]), t["weight"], t["bias"], t["r"], 1e-5, "lrelu", 0.2)).cpu()
        assert rel_err(ya, torch.nn.functional.leaky_relu(torch.from_numpy(g[f"{tag}.y"]), 0.2)) <= TOL, tag
        if tag == "c32":
            for dt, tol in (("f16", 2e-3), ("bf16", 1.5e-2)):
                x16 = t["x"].to(ops.TORCH_DT[dt]).contiguous(memory_format=torch.channels_last)
                y16 = ops.learnable_instance_norm(x16, t["weight"], t["bias"], t["r"])
                assert rel_err(y16.float().cpu(), g[f"{tag}.y"]) <= tol, (tag, dt)


def test_hvi_transform_reference_golden():
    """ie_rgb_to_hvi_nhwc_f32 / ie_hvi_to_rgb_nhwc_f32 against the reference class's own outputs (tests/golden/hvi_transform_*.npz).  Hue is discontinuous where two
    channels tie for the maximum and at the sector borders of the inverse: those few pixels may land on the other branch by rounding, everything else agrees to 1e-5."""
    g = np.load(os.path.join(GOLDEN, "hvi_transform_seed121.npz"))
    x = ops.to_nhwc(torch.from_numpy(g["image"]).cuda())
    for k, tag in ((0.2, "k20"), (0.35, "k35")):
        hvi = ops.to_nchw(ops.rgb_to_hvi(x, k)).cpu().numpy()
        d = np.abs(hvi - g[f"{tag}.hvi"]).max(1)
        assert (d > 1e-5).mean() <= 2e-3 and np.median(d) <= 1e-6, (tag, float(d.max()))
        for src, want in ((f"{tag}.hvi", f"{tag}.rgb"), (f"{tag}.pert", f"{tag}.rgb_pert")):
            rgb = ops.to_nchw(ops.hvi_to_rgb(ops.to_nhwc(torch.from_numpy(g[src]).cuda()), k)).cpu().numpy()
            d = np.abs(rgb - g[want]).max(1)
            assert (d > 1e-5).mean() <= 5e-3 and np.median(d) <= 1e-6, (src, float(d.max()))


def test_retinexnet_reference_golden():
    """The engine against RetinexNet.forward of the reference's own file run on CPU (tests/golden/retinexnet_c16_*.npz: DecomNet + RelightNet at channel = 16, their
    class bodies executed out of src/mon_extra/vision/enhance/llie/retinexnet/model.py by make_golden._extract): fp32 path inside 1e-4 on R, L, I_delta and S, and the
    training forward's two losses (forward_train_loss: DecomNet on both frames, RelightNet, nine terms, model.py:199-238)."""
    g = np.load(os.path.join(GOLDEN, "retinexnet_c16_seed81_2x42x58.npz"))
    sd = {k[len("w.decom."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.decom.")}
    sd.update({k[len("w.relight."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.relight.")})
    m = mon_amd.MODELS.build(name="retinexnet", channels=16, weights=dict(sd)).to("cuda")
    low, high = torch.from_numpy(g["low"]), torch.from_numpy(g["high"])
    out = m.forward({"image": low.cuda()})
    assert rel_err(out["reflectance"].cpu(), g["R_low"]) <= TOL
    assert rel_err(out["illumination"].cpu(), g["I_low_3"][:, :1]) <= TOL
    assert rel_err(out["illumination_delta"].cpu(), g["I_delta_3"][:, :1]) <= TOL
    assert rel_err(out["enhanced"].cpu(), g["S"]) <= TOL
    for phase in ("decom", "relight"):
        m.train_phase = phase
        tr = mon_amd.Trainer(m, optimizer={"name": "adam", "lr": 1e-4}, gradient_clip_val=None)
        res = tr.forward_loss({"image": low, "ref_image": high})
        assert abs(float(res["loss_decom"]) - float(g["loss_Decom"])) <= 2e-5 * float(g["loss_Decom"])
        assert abs(float(res["loss_relight"]) - float(g["loss_Relight"])) <= 2e-5 * float(g["loss_Relight"])
    for dt, tol in (("f16", 2e-3), ("bf16", 2e-2)):         # the 16-bit matrix-core path on the same golden (trained-like weights x 3: a harder case than the seed-6 test)
        m16 = mon_amd.MODELS.build(name="retinexnet", channels=16, weights=dict(sd), compute_dtype=dt).to("cuda")
        assert rel_err(m16.forward({"image": low.cuda()})["enhanced"].cpu(), g["S"]) <= tol, dt


def test_retinexnet_two_file_checkpoints(tmp_path):
    m = mon_amd.RetinexNet(seed=7)
    sd = m.state_dict()
    torch.save({k: v for k, v in sd.items() if k.startswith("net1_")}, tmp_path / "d.pt")
    torch.save({k: v for k, v in sd.items() if k.startswith("net2_")}, tmp_path / "r.pt")
    m2 = mon_amd.MODELS.build(name="retinexnet", weights={"decom": str(tmp_path / "d.pt"), "relight": str(tmp_path / "r.pt")})
    assert all(torch.equal(m2.state_dict()[k], sd[k]) for k in sd)


def test_building_blocks_against_torch():
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 24, 18, 22, generator=g)
    xn = ops.to_nhwc(x.cuda())
    assert torch.equal(ops.to_nchw(xn).cpu(), x)
    assert torch.equal(ops.to_nchw(ops.maxpool2x2(xn)).cpu(), F.max_pool2d(x, 2))
    assert torch.equal(ops.to_nchw(ops.maxpool2x2(ops.to_nhwc(x[:, :, :17, :21].contiguous().cuda()))).cpu(), F.max_pool2d(x[:, :, :17, :21], 2))
    for mode, kw in (("nearest", dict(mode="nearest")), ("bilinear", dict(mode="bilinear", align_corners=False)),
                     ("bilinear_align", dict(mode="bilinear", align_corners=True))):
        for size in ((36, 44), (9, 11), (25, 31)):
            ref = F.interpolate(x, size=size, **kw)
            assert (ops.to_nchw(ops.resize(xn, size, mode)).cpu() - ref).abs().max() <= 2e-6, (mode, size)
    assert torch.equal(ops.to_nchw(ops.pixel_shuffle(xn, 2)).cpu(), F.pixel_shuffle(x, 2))
    gamma, beta = torch.rand(24, generator=g) + 0.5, torch.randn(24, generator=g)
    ref = F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.2)
    got = ops.to_nchw(ops.instance_norm(xn, gamma.cuda(), beta.cuda(), 1e-5, "lrelu", 0.2)).cpu()
    assert (got - ref).abs().max() <= 2e-5
    # half instance norm (hinet.py:75-77): first half normalised, second half passed through
    ref = torch.cat([F.instance_norm(x[:, :12], weight=gamma[:12], bias=beta[:12], eps=1e-5), x[:, 12:]], 1)
    got = ops.to_nchw(ops.instance_norm(xn, gamma[:12].cuda(), beta[:12].cuda(), 1e-5, c_begin=0, c_count=12)).cpu()
    assert (got - ref).abs().max() <= 2e-5
    wt, bt = torch.randn(24, 10, 2, 2, generator=g) * 0.2, torch.randn(10, generator=g)
    ref = F.conv_transpose2d(x, wt, bt, stride=2)
    got = ops.to_nchw(ops.conv_transpose2x2(xn, ops.ConvT2x2Weights(wt, bt, "cuda"))).cpu()
    assert (got - ref).abs().max() <= 2e-5
    assert torch.equal(ops.to_nchw(ops.pad(xn, 1, 2, 0, 3)).cpu(), F.pad(x, [0, 3, 1, 2]))
    y = torch.randn(2, 24, 18, 22, generator=g)
    assert torch.equal(ops.to_nchw(ops.add(xn, ops.to_nhwc(y.cuda()))).cpu(), x + y)
    one = torch.randn(2, 1, 18, 22, generator=g)
    assert torch.equal(ops.to_nchw(ops.mul(xn, ops.to_nhwc(one.cuda()))).cpu(), x * one)
    assert torch.equal(ops.to_nchw(ops.clamp(xn, -0.5, 0.25)).cpu(), x.clamp(-0.5, 0.25))
    assert torch.equal(ops.to_nchw(ops.max_concat(xn)).cpu(), torch.cat([x.max(1, keepdim=True)[0], x], 1))
    assert torch.equal(ops.to_nchw(ops.mul_prefix(xn, ops.to_nhwc(one.cuda()))).cpu(), x[:, :23] * one)


@pytest.mark.parametrize("cfg", [dict(num_channels=16, hw=(32, 48), n=2), dict(num_channels=8, depth=3, in_pos_left=1, in_pos_right=1, hw=(20, 28), n=1),
                                 dict(num_channels=64, hw=(16, 32), n=1)], ids=["c16-d5", "c8-d3-hin1", "c64-default"])
def test_hinet_vs_oracle(cfg):
    """HINet (hinet.py:204-362): half instance norm, Conv4x4 stride 2, ConvTranspose 2x2, CSFF, SAM."""
    hw, n = cfg.pop("hw"), cfg.pop("n")
    m = mon_amd.MODELS.build(name="hinet_re", seed=5, **cfg).to("cuda")
    x = torch.rand(n, 3, *hw, generator=torch.Generator().manual_seed(4))
    out = m.forward({"image": x.cuda()})
    assert list(out.keys()) == ["stage1", "enhanced"]
    ref = og.hinet_forward(m.state_dict(), x, depth=cfg.get("depth", 5), in_pos_left=cfg.get("in_pos_left", 0), in_pos_right=cfg.get("in_pos_right", 4))
    for k in ("stage1", "enhanced"):
        assert rel_err(out[k].cpu(), ref[k]) <= TOL, k
    with pytest.raises(ValueError):
        m.forward({"image": torch.rand(1, 3, hw[0] + 2, hw[1]).cuda()})
    if cfg["num_channels"] == 16:                           # infer(): resize to a multiple of 32 and back (vision/model.py:97-127)
        y = torch.rand(1, 3, 40, 56, generator=torch.Generator().manual_seed(1))
        got = m.infer({"image": y.clone()})
        assert list(got.keys())[-2:] == ["enhanced", "time"] and tuple(got["enhanced"].shape) == (1, 3, 40, 56)
        yr = F.interpolate(y, size=(64, 64), mode="bilinear", align_corners=False)
        r = F.interpolate(og.hinet_forward(m.state_dict(), yr)["enhanced"], size=(40, 56), mode="bilinear", align_corners=False)
        assert rel_err(got["enhanced"].cpu(), r) <= TOL


# ---- 16-bit matrix-core path (compute_dtype f16 / bf16): same graphs, ie_mconv_nhwc + the 16-bit streaming ops ----------
TOL16 = {"f16": 1e-3, "bf16": 1.5e-2}         # fp16: the BASELINE bound itself (measured 3e-4 U-Net, 4e-4 RetinexNet; tools/gen16_diag.py); bf16 measured 2e-3 .. 6e-3


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("cfg", [dict(base_channels=32, num_classes=3, enhance=True, upsampler="bilinear", hw=(135, 70)),
                                 dict(base_channels=32, num_classes=3, enhance=True, upsampler="pixelshuffle", hw=(48, 64)),
                                 dict(base_channels=64, num_classes=34, enhance=False, upsampler="bilinear", hw=(33, 47))],
                         ids=["unet32-enh-odd", "unet32-pixelshuffle", "unet64-seg-odd"])
def test_unet_16bit_vs_oracle(cfg, dt):
    cfg = dict(cfg)
    hw = cfg.pop("hw")
    m = mon_amd.MODELS.build(name="unet", seed=3, compute_dtype=dt, **cfg).to("cuda")
    assert m.compute_dtype == dt
    x = torch.rand(2, 3, *hw, generator=torch.Generator().manual_seed(2))
    out = m.forward({"image": x.cuda()})
    key = "enhanced" if cfg["enhance"] else "semantic"
    ref = og.unet_forward(m.state_dict(), x, cfg["upsampler"], cfg["enhance"])[key]
    assert out[key].dtype == torch.float32 and tuple(out[key].shape) == tuple(ref.shape)
    # the fp16 enhancement nets sit inside BASELINE's 1e-3 bound on the matrix cores (measured 4.1e-4 / 3.2e-4, profiles/r02/gen16_parity.log)
    assert rel_err(out[key].cpu(), ref) <= (1e-3 if (dt == "f16" and cfg["enhance"]) else TOL16[dt])


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_retinexnet_16bit_vs_oracle(dt):
    m = mon_amd.MODELS.build(name="retinexnet", seed=3, compute_dtype=dt).to("cuda")
    x = torch.rand(2, 3, 45, 67, generator=torch.Generator().manual_seed(2))
    out = m.forward({"image": x.cuda()})
    ref = og.retinexnet_forward(m.state_dict(), x)
    assert list(out.keys()) == ["reflectance", "illumination", "illumination_delta", "enhanced"]
    for k in out:
        assert rel_err(out[k].cpu(), ref[k]) <= TOL16[dt], k


def test_compute_dtype_is_validated():
    with pytest.raises(ValueError):
        mon_amd.MODELS.build(name="unet", compute_dtype="int8")


def test_16bit_streaming_ops_against_torch():
    g = torch.Generator().manual_seed(0)
    for tdt, tol in ((torch.float16, 1e-3), (torch.bfloat16, 8e-3)):
        x = torch.randn(2, 16, 11, 14, generator=g).to(tdt)
        xd = x.cuda().contiguous(memory_format=torch.channels_last)
        xf = x.float()
        assert torch.equal(ops.maxpool2x2(xd).float().cpu(), F.max_pool2d(xf, 2))
        for mode, kw in (("nearest", dict(mode="nearest")), ("bilinear", dict(mode="bilinear", align_corners=False)),
                         ("bilinear_align", dict(mode="bilinear", align_corners=True))):
            assert rel_err(ops.resize(xd, (22, 28), mode).float().cpu(), F.interpolate(xf, size=(22, 28), **kw)) <= tol, mode
        assert torch.equal(ops.pad(xd, 1, 2, 0, 3).float().cpu(), F.pad(xf, [0, 3, 1, 2]))
        x4 = torch.randn(2, 32, 5, 6, generator=g).to(tdt)
        assert torch.equal(ops.pixel_shuffle(x4.cuda().contiguous(memory_format=torch.channels_last), 2).float().cpu(), F.pixel_shuffle(x4.float(), 2))
        y = torch.randn(2, 16, 11, 14, generator=g).to(tdt)
        yd = y.cuda().contiguous(memory_format=torch.channels_last)
        assert rel_err(ops.add(xd, yd).float().cpu(), xf + y.float()) <= tol
        assert rel_err(ops.mul(xd, yd).float().cpu(), xf * y.float()) <= tol
        assert rel_err(ops.mul(xd, yd[:, :1].contiguous(memory_format=torch.channels_last)).float().cpu(), xf * y.float()[:, :1]) <= tol
        gam, bet = torch.randn(8, generator=g), torch.randn(8, generator=g)
        ref = torch.cat([F.instance_norm(xf[:, :8], weight=gam, bias=bet, eps=1e-5), xf[:, 8:]], 1)
        got = ops.instance_norm(xd, gam.cuda(), bet.cuda(), 1e-5, "lrelu", 0.2, 0, 8)
        assert rel_err(got.float().cpu(), F.leaky_relu(ref, 0.2)) <= tol
        c = ops.cast(ops.to_nhwc(torch.rand(1, 3, 6, 7, generator=g).cuda()), tdt, 8)
        assert c.shape[1] == 8 and float(c[:, 3:].float().abs().max()) == 0.0
        back = ops.cast(c, torch.float32, 3)
        assert back.dtype == torch.float32 and back.shape[1] == 3


@pytest.mark.parametrize("dt,tol_ll,tol_hi", [("f16", 5e-3, 2e-3), ("bf16", 4e-2, 1.5e-2)])
def test_llunetpp_hinet_zero_ig_16bit_vs_oracle(dt, tol_ll, tol_hi):
    """Deep instance-norm networks with random weights amplify the 16-bit storage rounding (measured 1.8e-3 / 1.3e-2 on
    LLUnet++, tools/gen16_diag.py); the fp32 path of the same graphs is the 1e-4 parity path."""
    g = torch.Generator().manual_seed(2)
    x = torch.rand(1, 3, 32, 48, generator=g)
    m = mon_amd.MODELS.build(name="llunet++_re", seed=31, compute_dtype=dt).to("cuda")
    assert rel_err(m.forward({"image": x.cuda()})["enhanced"].cpu(), og.llunetpp_forward(m.state_dict(), x)["enhanced"]) <= tol_ll
    x = torch.rand(2, 3, 32, 48, generator=g)
    m = mon_amd.MODELS.build(name="hinet_re", seed=5, num_channels=16, compute_dtype=dt).to("cuda")
    out, ref = m.forward({"image": x.cuda()}), og.hinet_forward(m.state_dict(), x)
    assert list(out.keys()) == ["stage1", "enhanced"]
    for k in out:
        assert rel_err(out[k].cpu(), ref[k]) <= tol_hi, k
    m = mon_amd.MODELS.build(name="zero_ig_enhance", seed=5, compute_dtype=dt).to("cuda")
    assert rel_err(m.forward({"image": x.cuda()})["enhanced"].cpu(), og.zero_ig_enhance_forward(m.state_dict(), x)["enhanced"]) <= tol_hi
    with pytest.raises(capi_err()):
        mon_amd.MODELS.build(name="hinet_re", seed=5, num_channels=8, compute_dtype=dt).to("cuda").forward({"image": x.cuda()})


def capi_err():
    return mon_amd.capi.IEError


@pytest.mark.parametrize("dt,tol", [("f16", 2e-3), ("bf16", 1.5e-2)])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "zero_dcepp_*.npz"))), ids=lambda p: os.path.basename(p)[11:-4])
def test_zero_dcepp_16bit_golden(path, dt, tol):
    """Zero-DCE++ on the 16-bit path: depthwise 3x3 (ie_dwconv3x3_nhwc) + 1x1 matrix-core conv per layer, against the
    reference-generated goldens (measured up to 9e-4 / 7.3e-3 on the gain-8 golden weights)."""
    g = np.load(path)
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    m = mon_amd.MODELS.build(name="zero_dce++_re", scale_factor=float(g["scale_factor"]), weights=sd, compute_dtype=dt).to("cuda")
    out = m.forward({"image": torch.from_numpy(g["image"]).cuda()})
    assert list(out.keys()) == ["adjust", "enhanced"]
    assert rel_err(out["enhanced"].cpu(), g["enhanced"]) <= tol
    assert rel_err(out["adjust"].cpu(), g["adjust"]) <= tol


def test_depthwise3x3_16bit_against_torch():
    g = torch.Generator().manual_seed(4)
    for tdt, tol in ((torch.float16, 1e-3), (torch.bfloat16, 8e-3)):
        x = torch.randn(2, 24, 13, 19, generator=g).to(tdt)
        w = torch.randn(24, 1, 3, 3, generator=g) * 0.3
        b = torch.randn(24, generator=g)
        ref = F.relu(F.conv2d(x.float(), w, b, padding=1, groups=24))
        got = ops.depthwise3x3_16(x.cuda().contiguous(memory_format=torch.channels_last), w[:, 0].reshape(24, 9).t().contiguous().cuda(), b.cuda(), "relu")
        assert rel_err(got.float().cpu(), ref) <= tol


def test_capture_graph_is_bit_identical_and_shape_checked():
    """Serving path: forward captured into a HIP graph replays to the same bits as the eager launches."""
    g = torch.Generator().manual_seed(8)
    for cfg in (dict(name="unet", base_channels=32, num_classes=3, enhance=True, compute_dtype="bf16"), dict(name="zero_dce_re", compute_dtype="f16"),
                dict(name="retinexnet")):
        m = mon_amd.MODELS.build(seed=2, **cfg).to("cuda")
        x = torch.rand(2, 3, 48, 64, generator=g).cuda()
        ref = {k: v.clone() for k, v in m.forward({"image": x}).items()}
        run = m.capture_graph(x)
        y = torch.rand(2, 3, 48, 64, generator=g).cuda()
        ref_y = {k: v.clone() for k, v in m.forward({"image": y}).items()}
        out = run(x)
        assert list(out.keys()) == list(ref.keys())
        for k in ref:
            assert torch.equal(out[k], ref[k]), (cfg["name"], k)
        out = run(y)                                   # new input through the same graph
        for k in ref_y:
            assert torch.equal(out[k], ref_y[k]), (cfg["name"], k)
        with pytest.raises(ValueError):
            run(torch.rand(1, 3, 48, 64).cuda())


def test_compute_efficiency_score_matches_survey_counts():
    """vision/model.py:36-70.  MACs: Zero-DCE 79,200/px, U-Net-32 634.5 GFLOP/2 per 1080p frame -> checked at 512x512
    through the ratio, RetinexNet 1374.5/2 GMAC per 1080p frame."""
    zd = mon_amd.MODELS.build(name="zero_dce_re", seed=1).to("cuda")
    flops, params, t = zd.compute_efficiency_score(image_size=512, runs=3)
    assert flops == 79200 * 512 * 512 and params == 79416 and t > 0
    u = mon_amd.MODELS.build(name="unet", base_channels=32, num_classes=3, enhance=True, seed=1, compute_dtype="bf16").to("cuda")
    flops, params, t = u.compute_efficiency_score(image_size=512, runs=2)
    per_px_1080 = 634.5e9 / 2 / (1080 * 1920)
    assert abs(flops / (512 * 512) - per_px_1080) / per_px_1080 < 0.03          # 1080 rows pad at one level; 8-channel input padding
    r = mon_amd.MODELS.build(name="retinexnet", seed=1).to("cuda")
    flops, _, _ = r.compute_efficiency_score(image_size=256, runs=1)
    per_px = 1374.5e9 / 2 / (1080 * 1920)
    assert abs(flops / (256 * 256) - per_px) / per_px < 0.03


# ---- V-channel / guided curve models: colour space, box / guided filters, ZeroDCEV, GCENet ------------------------------
def _smooth(n, c, h, w, seed, lo=0.05, hi=0.9):
    """Textured but not white-noise test image: low-resolution noise up-sampled + a little fine noise (guided-filter coefficients
    A = cov / (var + eps) are ill-conditioned on flat windows, in the reference as much as here)."""
    g = torch.Generator().manual_seed(seed)
    base = F.interpolate(torch.rand(n, c, max(h // 6, 2), max(w // 6, 2), generator=g), size=(h, w), mode="bilinear", align_corners=False)
    return (lo + (hi - lo) * (0.85 * base + 0.15 * torch.rand(n, c, h, w, generator=g))).contiguous()


def test_filter_ops_against_oracle():
    from oracle import filters as of
    x = _smooth(2, 3, 37, 53, seed=1)
    xn = ops.to_nhwc(x.cuda())
    hsv, v = ops.rgb_to_hsv(xn)
    ref = of.rgb_to_hsv(x)
    assert rel_err(ops.to_nchw(hsv).cpu()[:, 1:], ref[:, 1:]) <= 1e-6 and torch.equal(ops.to_nchw(v).cpu(), ref[:, 2:3])
    dh = (ops.to_nchw(hsv).cpu()[:, 0] - ref[:, 0]).abs()
    assert float(torch.minimum(dh, 2 * np.pi - dh).max()) <= 1e-5                 # hue is an angle
    assert rel_err(ops.to_nchw(ops.hsv_to_rgb(hsv)).cpu(), x) <= 2e-6
    vnew = torch.rand(2, 1, 37, 53, generator=torch.Generator().manual_seed(2))
    h2 = ref.clone(); h2[:, 2:3] = vnew
    assert rel_err(ops.to_nchw(ops.hsv_to_rgb(ops.to_nhwc(ref.cuda()), ops.to_nhwc(vnew.cuda()))).cpu(), of.hsv_to_rgb(h2)) <= 2e-6
    for r in (1, 3):
        assert rel_err(ops.to_nchw(ops.box_filter(xn, r)).cpu(), of.box_filter(x, r)) <= 1e-5
    y = (x * 0.8 + 0.1 * _smooth(2, 3, 37, 53, seed=3)).contiguous()
    yn = ops.to_nhwc(y.cuda())
    assert rel_err(ops.to_nchw(ops.guided_filter(xn, yn, 3, 1e-4)).cpu(), of.guided_filter(x, y, 3, 1e-4)) <= 1e-3
    x1, y1 = x[:, :1].contiguous(), y[:, :1].contiguous()
    hr = _smooth(2, 1, 70, 91, seed=4)
    got = ops.fast_guided_filter(ops.to_nhwc(x1.cuda()), ops.to_nhwc(y1.cuda()), ops.to_nhwc(hr.cuda()), 1, 1e-4, clip=(0.0, 1.0))
    assert rel_err(ops.to_nchw(got).cpu(), of.fast_guided_filter(x1, y1, hr, 1, 1e-4).clip(0, 1)) <= 1e-3
    assert rel_err(ops.to_nchw(ops.resize_bicubic_align(xn, (70, 91))).cpu(), F.interpolate(x, (70, 91), mode="bicubic", align_corners=True)) <= 1e-5
    for k in (3, 9):
        assert torch.equal(ops.to_nchw(ops.median_blur(xn, k)).cpu(), of.median_blur(x, k))
    assert rel_err(ops.to_nchw(ops.sobel(xn, 1e-6)).cpu(), of.sobel(x, 1e-6)) <= 1e-6
    assert torch.equal(ops.to_nchw(ops.div_by_max(xn)).cpu(), x / x.max())
    d = _smooth(1, 1, 37, 53, seed=5)
    assert torch.equal(ops.to_nchw(ops.threshold_of_max(ops.sobel(ops.to_nhwc(d.cuda()), 1e-6), 0.3)).cpu(), (of.sobel(d) / of.sobel(d).max() > 0.3).float())
    adj = torch.tanh(torch.randn(2, 15, 37, 53, generator=torch.Generator().manual_seed(6)))
    yv = x1
    for r_ in torch.split(adj, 1, dim=1):
        yv = yv + r_ * (torch.pow(yv, 2) - yv)
    assert torch.equal(ops.to_nchw(ops.le_curve(ops.to_nhwc(x1.cuda()), ops.to_nhwc(adj.cuda()), 15)).cpu(), yv)


@pytest.mark.parametrize("dt,tol", [("f32", 1e-3), ("f16", 1e-2)])
def test_zero_dce_v_vs_oracle(dt, tol):
    """zero_dce_v against the restated reference flow.  1e-3, not 1e-4: the reference's box filter is a difference of fp32 cumulative sums
    (1.5e-5 absolute noise on a window sum) feeding A = cov / (var + 1e-8); the kernel sums each window directly in fp64."""
    from oracle import filters as of
    m = mon_amd.MODELS.build(name="zero_dce_v", seed=2, down_size=32, compute_dtype=dt).to("cuda")
    sd = {k: (v * 6 if k.endswith("weight") else v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    x = _smooth(2, 3, 45, 70, seed=7, lo=0.02, hi=0.5)
    out, ref = m.forward({"image": x.cuda()}), of.zero_dce_v_forward(sd, x, 15, 32)
    assert list(out.keys()) == ["image_v", "image_v_lr", "adjust", "image_v_fixed_lr", "image_v_fixed", "enhanced"]
    for k in ("image_v", "image_v_lr"):
        assert torch.equal(out[k].cpu(), ref[k]), k
    for k in ("adjust", "image_v_fixed_lr", "image_v_fixed", "enhanced"):
        assert tuple(out[k].shape) == tuple(ref[k].shape) and rel_err(out[k].cpu(), ref[k]) <= (tol if k != "adjust" or dt != "f32" else 1e-4), k
    assert float(out["enhanced"].max()) == 1.0
    with pytest.raises(ValueError):
        mon_amd.MODELS.build(name="zero_dce_v", in_channels=3)


def test_gcenet_vs_oracle():
    from oracle import filters as of
    m = mon_amd.MODELS.build(name="gcenet", seed=3, debug=True).to("cuda")
    sd = {k: (v * 8 if k.endswith("weight") else v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    x = _smooth(1, 3, 48, 72, seed=8, lo=0.02, hi=0.6)
    depth = _smooth(1, 1, 48, 72, seed=9)
    out, ref = m.forward({"image": x.cuda(), "depth": depth.cuda()}), of.gcenet_forward(sd, x, depth)
    assert list(out.keys()) == ["adjust", "depth", "edge", "bam", "enhanced"]
    assert torch.equal(out["edge"].cpu(), ref["edge"]) and 0.02 < float(ref["edge"].mean()) < 0.98
    assert rel_err(out["bam"].cpu(), ref["bam"]) <= 1e-6 and rel_err(out["adjust"].cpu(), ref["adjust"]) <= 1e-4
    assert rel_err(out["enhanced"].cpu(), ref["enhanced"]) <= 1e-3
    plain = mon_amd.MODELS.build(name="gcenet", weights=sd).to("cuda")
    assert list(plain.forward({"image": x.cuda(), "depth": depth.cuda()}).keys()) == ["enhanced"]
    with pytest.raises(ValueError):
        plain.forward({"image": x.cuda()})                       # use_depth=True needs the depth map
    nodepth = mon_amd.MODELS.build(name="gcenet", seed=3, use_depth=False, bam_gamma=0.0).to("cuda")
    got = nodepth.forward({"image": x.cuda()})["enhanced"].cpu()
    want = of.gcenet_forward(nodepth.state_dict(), x, None, use_depth=False, bam_gamma=0.0)["enhanced"]
    assert rel_err(got, want) <= 1e-3


def test_gcenet_registry_variants_resolve_and_predict():
    """`gcenet_zsn2n` / `gcenet_instance` (gcenet.py:504, 539): the names resolve, load GCENet checkpoints and predict like GCENet (their forward IS GCENet's);
    the parts that need a GCENet training step say so instead of silently doing something else."""
    base = mon_amd.MODELS.build(name="gcenet", seed=3).to("cuda")
    sd = base.state_dict()
    x = _smooth(1, 3, 32, 48, seed=8, lo=0.02, hi=0.6)
    depth = _smooth(1, 1, 32, 48, seed=9)
    want = base.forward({"image": x.cuda(), "depth": depth.cuda()})["enhanced"]
    for name in ("gcenet_zsn2n", "gcenet_instance"):
        m = mon_amd.MODELS.build(name=name, weights=dict(sd)).to("cuda")
        assert m.name == name and torch.equal(m.forward({"image": x.cuda(), "depth": depth.cuda()})["enhanced"], want)
    with pytest.raises(mon_amd.capi.IEError):
        mon_amd.MODELS.build(name="gcenet_zsn2n", weights=dict(sd)).to("cuda").forward_loss({"image": x.cuda(), "depth": depth.cuda()})
    inst = mon_amd.MODELS.build(name="gcenet_instance", weights=dict(sd)).to("cuda")
    with pytest.raises(mon_amd.capi.IEError):
        inst.infer({"image": x.cuda(), "depth": depth.cuda()})
    assert "enhanced" in inst.infer({"image": x.cuda(), "depth": depth.cuda()}, epochs=0)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_gcenet_16bit_vs_oracle(dt):
    """GCENet with EnhanceNet on the matrix cores (16-bit activations, fp32 accumulation): the curve map within the 16-bit bound of the oracle's, the
    enhanced image behind 15 curve iterations and the guided filter likewise; the priors (edge, bam) are the fp32 path's, bit for bit."""
    from oracle import filters as of
    m = mon_amd.MODELS.build(name="gcenet", seed=3, debug=True, compute_dtype=dt).to("cuda")
    sd = {k: (v * 8 if k.endswith("weight") else v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    x = _smooth(2, 3, 48, 72, seed=8, lo=0.02, hi=0.6)
    depth = _smooth(2, 1, 48, 72, seed=9)
    out, ref = m.forward({"image": x.cuda(), "depth": depth.cuda()}), of.gcenet_forward(sd, x, depth)
    assert torch.equal(out["edge"].cpu(), ref["edge"]) and rel_err(out["bam"].cpu(), ref["bam"]) <= 1e-6
    tol = 2e-3 if dt == "f16" else 1.5e-2
    assert rel_err(out["adjust"].cpu(), ref["adjust"]) <= tol
    assert rel_err(out["enhanced"].cpu(), ref["enhanced"]) <= tol
    m32 = mon_amd.MODELS.build(name="gcenet", weights=sd, debug=True).to("cuda")
    assert rel_err(out["adjust"].cpu(), m32.forward({"image": x.cuda(), "depth": depth.cuda()})["adjust"].cpu()) <= tol
    nodepth = mon_amd.MODELS.build(name="gcenet", seed=3, use_depth=False, use_edge=False, bam_gamma=0.0, compute_dtype=dt).to("cuda")
    got = nodepth.forward({"image": x.cuda()})["enhanced"].cpu()
    want = of.gcenet_forward(nodepth.state_dict(), x, None, use_depth=False, use_edge=False, bam_gamma=0.0)["enhanced"]
    assert rel_err(got, want) <= tol


# ---- BASELINE.json configs[2] / configs[3] at their frame size: properties that need no CPU reference ---------------------------------
@pytest.mark.parametrize("cfg", [dict(name="unet", base_channels=32, num_classes=3, enhance=True, upsampler="pixelshuffle"),
                                 dict(name="unet", base_channels=32, num_classes=3, enhance=True),
                                 dict(name="retinexnet")], ids=["unet32-pixelshuffle", "unet32-bilinear", "retinexnet"])
def test_1080p_properties_unet32_retinexnet(cfg):
    """1920x1080 frames through the bf16 matrix-core path (the configuration bench.py's extras time): determinism, batch
    independence, agreement of a crop on its interior (U-Net's receptive field is larger than any practical crop, so that check
    runs on RetinexNet's DecomNet half only), output range, and agreement with the fp32 parity path of the SAME graph on one frame
    within the 16-bit bound of the small-size oracle tests."""
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(2, 3, 1080, 1920, generator=g) ** 2.2 * 0.5).cuda()
    m = mon_amd.MODELS.build(seed=3, compute_dtype="bf16", **cfg).to("cuda")
    a = m.forward({"image": x})
    b = m.forward({"image": x})
    for k in a:
        assert torch.equal(a[k], b[k]), k                                        # deterministic
    one = m.forward({"image": x[1:2]})
    for k in a:
        assert torch.equal(a[k][1:2], one[k]), k                                 # frames are independent
    e = a["enhanced"]
    assert tuple(e.shape) == (2, 3, 1080, 1920) and bool(torch.isfinite(e).all())
    m32 = mon_amd.MODELS.build(seed=3, compute_dtype="f32", **cfg).to("cuda")
    ref = m32.forward({"image": x[:1]})["enhanced"]
    assert rel_err(e[:1].cpu(), ref.cpu()) <= 1.5e-2                             # TOL16["bf16"] of the oracle tests above
    if cfg["name"] == "unet":
        assert float(e.min()) >= 0.0 and float(e.max()) <= 1.0                   # enhance=True clamps (llunetpp.py:241-242)
    else:
        # DecomNet: 7 convolutions, receptive radius 4 + 6 = 10 (replicate padding only touches the border): a crop's reflectance
        # agrees with the full frame's away from its edges
        crop = m.forward({"image": x[:1, :, 400:560, 700:1000].contiguous()})["reflectance"]
        assert torch.equal(crop[:, :, 10:-10, 10:-10], a["reflectance"][:1, :, 410:550, 710:990])


def test_lllinet_hvi_vs_oracle():
    """`lllinet_hvi` (lllinet.py:262-388): the HVI colour transform pair (core/image/color.py:313-434) against its restatement -- grey pixels, black pixels,
    every hue sector, out-of-range inputs of the inverse -- and the model (transform -> LLLINet grid -> inverse -> clamp) on both execution paths."""
    g = torch.Generator().manual_seed(21)
    x = torch.rand(2, 3, 24, 40, generator=g)
    x[0, :, 0, :8] = x[0, :1, 0, :8]                 # grey pixels: min == max
    x[0, :, 1, :8] = 0.0                             # black pixels
    x[1, 0, 2, :8] = 1.0                             # R is the maximum (the `% 6` branch, negative hue before the modulo where G < B)
    hvi = ops.to_nchw(ops.rgb_to_hvi(ops.to_nhwc(x.cuda()), 0.2)).cpu()
    assert (hvi - og.rgb_to_hvi(x, 0.2)).abs().max() <= 2e-6
    y = torch.randn(2, 3, 24, 40, generator=g) * 0.8
    y[:, 2] = y[:, 2].abs()
    got = ops.to_nchw(ops.hvi_to_rgb(ops.to_nhwc(y.cuda()), 0.2)).cpu()
    want = og.hvi_to_rgb(y, 0.2)
    # the inverse is discontinuous where h * 6 crosses an integer: compare away from those pixels (1 ulp of atan2 decides the sector there)
    pi = 3.141592653589793
    Hc, Vc, v = y[:, 0].clamp(-1, 1), y[:, 1].clamp(-1, 1), y[:, 2].clamp(0, 1)
    cs = ((v * 0.5 * pi).sin() + 1e-8).pow(0.2)
    h6 = ((torch.atan2((Vc / (cs + 1e-8)).clamp(-1, 1), (Hc / (cs + 1e-8)).clamp(-1, 1)) / (2 * pi)) % 1) * 6
    safe = ((h6 - h6.round()).abs() > 1e-4).unsqueeze(1).expand_as(got)
    assert float(safe.float().mean()) > 0.99 and (got - want)[safe].abs().max() <= 5e-6
    assert torch.equal(ops.to_nchw(ops.hvi_to_rgb(ops.to_nhwc(y.cuda()), 0.2, clamp01=True)).cpu(), got.clamp(0, 1))
    # round trip on in-gamut colours
    back = ops.to_nchw(ops.hvi_to_rgb(ops.rgb_to_hvi(ops.to_nhwc(x.cuda()), 0.2), 0.2)).cpu()
    assert (back - x).abs().max() <= 2e-5
    m = mon_amd.MODELS.build(name="lllinet_hvi", seed=9).to("cuda")
    assert m.name == "lllinet_hvi" and "trans.density_k" in m.state_dict()
    sd = m.state_dict()
    img = _smooth(1, 3, 32, 48, seed=4, lo=0.05, hi=0.9)
    ref = og.lllinet_hvi_forward(sd, img)["enhanced"]
    out = m.forward({"image": img.cuda()})["enhanced"].cpu()
    assert tuple(out.shape) == (1, 3, 32, 48) and float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    assert rel_err(out, ref) <= 1e-3
    sd2 = dict(sd); sd2["trans.density_k"] = torch.full([1], 0.35)
    m2 = mon_amd.MODELS.build(name="lllinet_hvi", weights=sd2).to("cuda")
    assert rel_err(m2.forward({"image": img.cuda()})["enhanced"].cpu(), og.lllinet_hvi_forward(sd2, img)["enhanced"]) <= 1e-3
    m16 = mon_amd.MODELS.build(name="lllinet_hvi", weights=dict(sd), compute_dtype="f16").to("cuda")
    assert rel_err(m16.forward({"image": img.cuda()})["enhanced"].cpu(), ref) <= 3e-2


def test_lllinet_vs_oracle_and_stat_layers():
    """LLLINet (learnable instance norm, SimAM, dense UNet++ wiring) against the restated reference forward; the statistics layers
    on their own against torch; the channel-attention pieces (global average pool + per-channel rescale, mprnet.py:51-82)."""
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 24, 19, 23, generator=g)
    xn = ops.to_nhwc(x.cuda())
    gam, bet, r = torch.rand(24, generator=g) + 0.5, torch.randn(24, generator=g), torch.rand(24, generator=g)
    got = ops.to_nchw(ops.learnable_instance_norm(xn, gam.cuda(), bet.cuda(), r.cuda(), 1e-5, "lrelu", 0.2)).cpu()
    assert (got - F.leaky_relu(og.learnable_instance_norm(x, gam, bet, r), 0.2)).abs().max() <= 2e-5
    assert (ops.to_nchw(ops.simam(xn, 1e-4)).cpu() - og.simam(x)).abs().max() <= 2e-5
    assert (ops.to_nchw(ops.global_avg_pool(xn)).cpu() - x.mean((2, 3), keepdim=True)).abs().max() <= 1e-6
    sc = torch.rand(2, 24, generator=g)
    assert torch.equal(ops.to_nchw(ops.channel_affine(xn, sc.cuda())).cpu(), x * sc[:, :, None, None])
    assert (ops.to_nchw(ops.channel_affine(xn, sc.cuda(), residual=ops.to_nhwc((x * 0.5).cuda()))).cpu() - (x * sc[:, :, None, None] + x * 0.5)).abs().max() <= 1e-6
    m = mon_amd.MODELS.build(name="lllinet", seed=9).to("cuda")
    sd = m.state_dict()
    for k in sd:                                                        # non-trivial norm parameters and blend ratios
        if k.endswith("norm1.r"):
            sd[k] = torch.rand(sd[k].shape, generator=g)
        elif k.endswith("norm1.weight"):
            sd[k] = 1.0 + 0.2 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith("norm1.bias"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    sd["final.bias"] = sd["final.bias"] + 0.45
    m.load_state_dict(sd)
    assert "conv0_0.norm1.r" not in sd and "conv1_0.norm1.r" in sd and tuple(sd["conv0_3.conv3.weight"].shape) == (32, 2 * 224, 3, 3)
    img = torch.rand(1, 3, 64, 96, generator=g)
    out = m.forward({"image": img.cuda()})["enhanced"].cpu()
    ref = og.lllinet_forward(sd, img)["enhanced"]
    assert rel_err(out, ref) <= 5e-4 and 0.05 < float(ref.mean()) < 0.95
    for dt, tol in (("f16", 4e-3), ("bf16", 4e-2)):                     # 16-bit matrix-core path (RGB as 8 channels; statistics in fp64)
        m16 = mon_amd.MODELS.build(name="lllinet", weights=dict(sd), compute_dtype=dt).to("cuda")
        o16 = m16.forward({"image": img.cuda()})["enhanced"]
        assert o16.dtype == torch.float32 and rel_err(o16.cpu(), ref) <= tol, (dt, rel_err(o16.cpu(), ref))
    with pytest.raises(ValueError):
        m.forward({"image": torch.rand(1, 3, 40, 48).cuda()})


def test_mprnet_vs_oracle_and_patch_ops():
    """MPRNet (three stages, 4 / 2 / 1 patch hierarchy, channel attention, SAM, CSFF, ORSNet) against the restated reference forward at a
    reduced width; the patch plumbing (crop / spatial concat) against torch slicing."""
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 5, 12, 20, generator=g)
    xn = ops.to_nhwc(x.cuda())
    a, b = ops.crop(xn, 0, 0, 12, 8), ops.crop(xn, 0, 8, 12, 12)
    assert torch.equal(ops.to_nchw(a).cpu(), x[:, :, :, :8]) and torch.equal(ops.to_nchw(b).cpu(), x[:, :, :, 8:])
    assert torch.equal(ops.to_nchw(ops.cat_spatial([a, b], 3)).cpu(), x)
    t, u = ops.crop(xn, 0, 0, 5, 20), ops.crop(xn, 5, 0, 7, 20)
    assert torch.equal(ops.to_nchw(ops.cat_spatial([t, u], 2)).cpu(), x)
    kw = dict(num_channels=16, scale_unetfeats=8, scale_orsnetfeats=8, num_cab=2, reduction=4)
    m = mon_amd.MODELS.build(name="mprnet", seed=17, **kw).to("cuda")
    sd = m.state_dict()
    assert "stage2_encoder.csff_dec3.weight" in sd and "stage1_encoder.csff_enc1.weight" not in sd
    assert tuple(sd["stage3_orsnet.up_enc2.0.up.1.weight"].shape) == (24, 32, 1, 1) and "tail.bias" not in sd
    for k in sd:                       # the shared PReLU slope, away from its default; livelier gates
        if k.endswith("body.1.weight"):
            sd[k] = torch.full((1,), 0.1)
        elif k.endswith("conv_du.2.weight") or k.endswith("conv3.weight"):
            sd[k] = sd[k] * 3.0
    m.load_state_dict(sd)
    img = torch.rand(2, 3, 32, 48, generator=g)
    out = m.forward({"image": img.cuda()})
    ref = og.mprnet_forward(sd, img, num_cab=2)
    assert list(out) == ["stage1", "stage2", "enhanced"]
    for k in ref:
        assert tuple(out[k].shape) == tuple(ref[k].shape) and rel_err(out[k].cpu(), ref[k]) <= 2e-5, k
    assert float((ref["enhanced"] - img).abs().max()) > 1e-2 and float((ref["stage1"] - ref["stage2"]).abs().max()) > 1e-3
    mb = mon_amd.MODELS.build(name="mprnet", seed=3, bias=True, **kw).to("cuda")         # bias=True variant of every conv() / 1x1
    sdb = mb.state_dict()
    assert "tail.bias" in sdb and "stage3_orsnet.orb1.body.2.bias" not in sdb and "stage1_decoder.up21.up.1.bias" not in sdb
    outb = mb.forward({"image": img.cuda()})["enhanced"].cpu()
    assert rel_err(outb, og.mprnet_forward(sdb, img, num_cab=2)["enhanced"]) <= 2e-5
    with pytest.raises(ValueError):
        m.forward({"image": torch.rand(1, 3, 36, 48).cuda()})
    for dt, tol in (("f16", 4e-3), ("bf16", 3e-2)):                                   # 16-bit matrix-core path (images as 8 channels)
        m16 = mon_amd.MODELS.build(name="mprnet", weights=dict(sd), compute_dtype=dt, **kw).to("cuda")
        o16 = m16.forward({"image": img.cuda()})
        for k in ref:
            assert o16[k].dtype == torch.float32 and tuple(o16[k].shape) == tuple(ref[k].shape)
            assert rel_err(o16[k].cpu(), ref[k]) <= tol, (dt, k, rel_err(o16[k].cpu(), ref[k]))


def test_guided_filter_reference_golden():
    """ie_box_filter_nhwc_f32 / ie_guided_filter_nhwc_f32 against box_filter / BoxFilter / GuidedFilter of the reference's own files (tests/golden/guided_filter_*.npz):
    the class's window is ``radius`` pixels wide (ops.guided_filter_module), radius 1 returns the guide."""
    g = np.load(os.path.join(GOLDEN, "guided_filter_seed161.npz"))
    x, y = ops.to_nhwc(torch.from_numpy(g["x"]).cuda()), ops.to_nhwc(torch.from_numpy(g["y"]).cuda())
    for r in (1, 3):
        assert rel_err(ops.to_nchw(ops.box_filter(x, r)).cpu(), g[f"box_r{r}"]) <= 1e-5
    assert rel_err(ops.to_nchw(ops.box_filter(y, 2)).cpu(), g["box_k5"]) <= 1e-5
    for r, eps, tol in ((3, 1e-4, 1e-4), (1, 1e-2, 1e-5), (5, 1e-8, 5e-3)):
        got = ops.to_nchw(ops.guided_filter_module(x, y, r, eps)).cpu().numpy()
        assert np.abs(got - g[f"gf_r{r}_eps{eps:g}"]).max() <= tol, (r, eps, float(np.abs(got - g[f"gf_r{r}_eps{eps:g}"]).max()))
    x1 = ops.to_nhwc(torch.from_numpy(g["x"][:, :1].copy()).cuda())
    assert np.abs(ops.to_nchw(ops.guided_filter_module(x1, y, 3, 1e-4)).cpu().numpy() - g["gf_r3_gray_guide3"]).max() <= 1e-4


def test_lllinet_reference_blocks_golden():
    """The engine's LLLINet (``filters`` = the fixture's widths) against the reference's own UNetConvBlock / LearnableInstanceNorm2d / SimAM run on CPU
    (tests/golden/lllinet_w2_*.npz, grid order of lllinet.py:229-260): fp32 path."""
    g = np.load(os.path.join(GOLDEN, "lllinet_w2_seed151_2x32x48.npz"))
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    m = mon_amd.MODELS.build(name="lllinet", weights=sd, filters=[2, 4, 8, 16, 32]).to("cuda")
    out = m.forward({"image": torch.from_numpy(g["image"]).cuda()})["enhanced"].cpu()
    assert float(np.abs(out.numpy() - g["enhanced"]).max()) <= 5e-5
    with pytest.raises(ValueError):
        mon_amd.MODELS.build(name="lllinet", filters=[4, 8])


def test_mprnet_reference_blocks_golden():
    """The engine's MPRNet against the reference's own blocks run on CPU (tests/golden/mprnet_c8_*.npz: conv / CAB / SAM / Encoder / Decoder / ORSNet executed out of
    mprnet.py and called in MPRNet.forward's order): fp32 path on the three outputs."""
    g = np.load(os.path.join(GOLDEN, "mprnet_c8_seed141_2x24x32.npz"))
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    m = mon_amd.MODELS.build(name="mprnet", weights=sd, num_channels=8, scale_unetfeats=4, scale_orsnetfeats=4, num_cab=2, reduction=4).to("cuda")
    out = m.forward({"image": torch.from_numpy(g["image"]).cuda()})
    assert list(out) == ["stage1", "stage2", "enhanced"]
    for k in out:
        assert rel_err(out[k].cpu(), g[k]) <= 5e-5, (k, rel_err(out[k].cpu(), g[k]))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_channel_stats_and_affine_16bit(dt):
    """The 16-bit forms of the channel-attention leaves (vectorised 8 channels per thread, and the scalar fallback for C % 8 != 0)."""
    g = torch.Generator().manual_seed(33)
    for c, h, w in ((96, 37, 53), (8, 5, 3), (12, 9, 11)):
        x = torch.randn(3, c, h, w, generator=g).to(dt)
        xd = x.cuda().contiguous(memory_format=torch.channels_last)
        st = ops.channel_stats(xd).cpu()
        xf = x.double()
        assert (st[:, :, 0] - xf.sum((2, 3))).abs().max() <= 1e-3 * h * w ** 0.5
        assert (st[:, :, 1] - (xf * xf).sum((2, 3))).abs().max() <= 2e-3 * h * w
        assert (ops.to_nchw(ops.global_avg_pool(xd)).cpu()[:, :, 0, 0] - xf.mean((2, 3)).float()).abs().max() <= 1e-5
        sc, sh = torch.rand(3, c, generator=g) + 0.5, torch.randn(3, c, generator=g)
        if h * w > 1:
            sm = ops.simam(xd, 1e-4)
            assert sm.dtype == dt and rel_err(sm.float().cpu(), og.simam(x.float())) <= (1.5e-3 if dt == torch.float16 else 1e-2)
        y = ops.channel_affine(xd, sc.cuda(), sh.cuda(), "lrelu", 0.2)
        ref = F.leaky_relu(x.float() * sc[:, :, None, None] + sh[:, :, None, None], 0.2)
        assert y.dtype == dt and rel_err(y.float().cpu(), ref) <= (1.5e-3 if dt == torch.float16 else 1e-2)
        # the gate and the skip connection of a channel-attention block in one pass (mprnet.py:156-161): exactly the fp32 sum, rounded once
        rs = torch.randn(3, c, h, w, generator=g).to(dt)
        y2 = ops.channel_affine(xd, sc.cuda(), None, residual=rs.cuda().contiguous(memory_format=torch.channels_last))
        assert y2.dtype == dt and torch.equal(y2.float().cpu(), (x.float() * sc[:, :, None, None] + rs.float()).to(dt).float())
