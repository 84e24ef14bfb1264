"""CPU: oracle restatements of the other generators against reference-generated goldens
(Zero-DCE++, LLUnet++) and host-side contracts of the generator classes."""
import ast
import glob
import os

import numpy as np
import pytest
import torch

import mon_amd
from oracle import generators as og
from tests.helpers import GOLDEN, seeded_bn_state_dict, seeded_state_dict


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "zero_dcepp_*.npz"))), ids=lambda p: os.path.basename(p)[11:-4])
def test_zero_dcepp_restatement_matches_reference(path):
    g = np.load(path)
    sd_ref = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    # reference (original repo) key names -> mon names through the model's own key mapping
    m = mon_amd.ZeroDCEpp_RE(scale_factor=float(g["scale_factor"]), weights=dict(sd_ref))
    sd = m.state_dict()
    assert "e_conv3.dw_conv.weight" in sd and len(sd) == 28
    out = og.zero_dcepp_forward(sd, torch.from_numpy(g["image"]), scale_factor=float(g["scale_factor"]))
    assert np.abs(out["enhanced"].numpy() - g["enhanced"]).max() <= 1e-6
    assert np.abs(out["adjust"].numpy() - g["adjust"]).max() <= 1e-6


def _llunet_golden():
    g = np.load(os.path.join(GOLDEN, "llunetpp_seed31_1x32x48.npz"))
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["key_order"], g["key_shapes"])}
    return g, seeded_state_dict(shapes, int(g["seed"]))


def test_llunetpp_restatement_matches_reference():
    g, sd_extra = _llunet_golden()
    m = mon_amd.LLUnetpp_RE(weights=dict(sd_extra))            # original key spelling -> mon spelling
    sd = m.state_dict()
    assert m.params == 29048873 and "conv0_4.conv2_3.weight" in sd
    assert set(og.llunetpp_to_extra_keys(sd)) == set(sd_extra)
    out = og.llunetpp_forward(sd, torch.from_numpy(g["image"]))["enhanced"].numpy()
    assert np.abs(out - g["enhanced"]).max() <= 2e-5
    assert 0.2 < out.mean() < 0.6 and (out > 0).mean() > 0.8      # not a degenerate all-clamped output


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "hinet_*.npz"))), ids=lambda p: os.path.basename(p)[6:-4])
def test_hinet_restatement_matches_reference(path):
    """oracle.hinet_forward against the outputs of the reference's own hinet_arch.HINet (both stages)."""
    g = np.load(path)
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["key_order"], g["key_shapes"])}
    sd = seeded_state_dict(shapes, int(g["seed"]))
    m = mon_amd.MODELS.build(name="hinet_re", num_channels=int(g["wf"]), weights=dict(sd))
    assert set(m.state_dict()) == set(sd)                        # same checkpoint keys as the reference module
    out = og.hinet_forward(m.state_dict(), torch.from_numpy(g["image"]))
    assert np.abs(out["stage1"].numpy() - g["stage1"]).max() <= 2e-5
    assert np.abs(out["enhanced"].numpy() - g["enhanced"]).max() <= 2e-5


def _zero_ig_golden():
    g = np.load(os.path.join(GOLDEN, "zero_ig_seed51_2x40x56.npz"))
    return g, {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}


def test_zero_ig_restatement_matches_reference():
    """Enhance, Denoise1/2 and the predicting forward against the reference's Enhancer / Denoise_1 / Denoise_2 /
    Finetunemodel.forward; the checkpoint carries the original repository's key names (denoise_1 / denoise_2)."""
    g, sd_ref = _zero_ig_golden()
    m = mon_amd.MODELS.build(name="zero_ig_re", weights=dict(sd_ref))
    sd = m.state_dict()
    assert "denoise1.conv1.weight" in sd and "enhance.conv.1.running_var" in sd
    enh = {k[8:]: v for k, v in sd.items() if k.startswith("enhance.")}
    xe = torch.from_numpy(g["enhancer_in"])
    assert np.abs(og.zero_ig_enhance_forward(enh, xe)["enhanced"].numpy() - g["enhancer_out"]).max() <= 1e-6
    assert np.abs(og.zero_ig_denoise_forward(sd, xe, "denoise1").numpy() - g["denoise1_out"]).max() <= 1e-6
    assert np.abs(og.zero_ig_denoise_forward(sd, torch.cat([xe, xe.flip(1)], 1), "denoise2").numpy() - g["denoise2_out"]).max() <= 1e-6
    out = og.zero_ig_forward(sd, torch.from_numpy(g["image"]))
    assert list(out) == ["denoise", "enhanced"]
    assert np.abs(out["enhanced"].numpy() - g["enhanced"]).max() <= 2e-6
    assert np.abs(out["denoise"].numpy() - g["denoise"]).max() <= 2e-6
    assert 0.1 < float(g["enhanced"].mean()) < 0.6               # not a degenerate all-clamped output


def test_generator_registry_and_param_counts():
    counts = {"zero_dce++_re": 10561, "llunet++_re": 29048873, "unet": 17273954, "retinexnet": 555205}
    for name, n in counts.items():
        assert mon_amd.MODELS.build(name=name, seed=0).params == n, name
    u32 = mon_amd.MODELS.build(name="unet", seed=0, base_channels=32, num_classes=3, enhance=True)
    assert abs(u32.params - 4.32e6) < 0.02e6
    with pytest.raises(ValueError):
        mon_amd.UNet(upsampler="nearest")
    with pytest.raises(mon_amd.capi.IEError):
        u32.forward({"image": torch.rand(1, 3, 16, 16)})
    with pytest.raises(ValueError):
        u32.forward({"x": torch.rand(1, 3, 16, 16)})


def test_bn_folding_is_exact_in_fp64():
    from mon_amd.ops import fold_bn
    g = torch.Generator().manual_seed(0)
    w, b = torch.randn(8, 4, 3, 3, generator=g), torch.randn(8, generator=g)
    gamma, beta, mean, var = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g), torch.randn(8, generator=g), torch.rand(8, generator=g) + 0.5
    x = torch.randn(2, 4, 9, 9, generator=g)
    import torch.nn.functional as F
    ref = F.batch_norm(F.conv2d(x, w, b, padding=1), mean, var, gamma, beta, False, 0.1, 1e-5)
    w2, b2 = fold_bn(w, b, gamma, beta, mean, var)
    assert (F.conv2d(x, w2, b2, padding=1) - ref).abs().max() < 2e-5


def test_hinet_structure_and_oracle_runs():
    """hinet.py:204-304: 88.67 M parameters at the default width (the figure the HINet authors publish), reference
    state_dict key names, and the CPU restatement runs the two stages with CSFF + SAM."""
    m = mon_amd.MODELS.build(name="hinet_re", seed=0)
    assert m.params == 88_669_702
    keys = set(m.state_dict())
    for k in ("conv_01.weight", "down_path_1.0.norm.weight", "down_path_1.3.downsample.weight", "down_path_2.0.csff_enc.bias",
              "up_path_1.0.up.weight", "up_path_2.3.conv_block.identity.weight", "skip_conv_2.1.bias", "sam12.conv3.weight",
              "cat12.weight", "last.bias"):
        assert k in keys, k
    assert "down_path_1.4.downsample.weight" not in keys and "down_path_1.0.csff_enc.weight" not in keys
    assert tuple(m.state_dict()["up_path_1.0.up.weight"].shape) == (1024, 512, 2, 2)
    small = mon_amd.MODELS.build(name="hinet_re", num_channels=8, seed=1)
    x = torch.rand(1, 3, 32, 48, generator=torch.Generator().manual_seed(0))
    out = og.hinet_forward(small.state_dict(), x)
    assert list(out) == ["stage1", "enhanced"] and out["enhanced"].shape == x.shape
    with pytest.raises(mon_amd.capi.IEError):
        small.forward({"image": x})            # no CPU execution path


def _unet_golden(path):
    g = np.load(path)
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["key_order"], g["key_shapes"])}
    return g, seeded_bn_state_dict(shapes, int(g["seed"])), "bilinear" if bool(g["bilinear"]) else "convtranspose"


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "unet_mtfe_*.npz"))), ids=lambda p: os.path.basename(p)[10:-4])
def test_unet_restatement_matches_reference(path):
    """oracle.unet_forward (unet.py:35-197) against the outputs of the reference tree's own runnable U-Net of the same topology
    (mon_extra/.../mtfe/unet_model.py + unet_parts.py): both decoders, odd frames (decoder zero pad), eval BatchNorm, conv biases."""
    g, sd_ref, upsampler = _unet_golden(path)
    assert len(glob.glob(os.path.join(GOLDEN, "unet_mtfe_*.npz"))) == 2
    m = mon_amd.MODELS.build(name="unet", base_channels=16, num_classes=3, conv_bias=True, upsampler=upsampler, weights=dict(sd_ref))
    sd = m.state_dict()                                          # original key spelling -> mon spelling through the model's own mapping
    assert "down3.conv.norm2.running_var" in sd and "inc.conv1.bias" in sd
    assert set(sd) == set(og.unet_from_mtfe_keys(sd_ref))
    assert m.params == sum(v.numel() for v in sd_ref.values() if v.dim() > 0)   # (counts the BatchNorm buffers too)
    out = og.unet_forward(sd, torch.from_numpy(g["image"]), upsampler, taps=True)
    assert np.abs(out["inc"].numpy() - g["inc"]).max() <= 1e-6
    assert np.abs(out["bottleneck"].numpy() - g["bottleneck"]).max() <= 1e-6
    assert np.abs(out["semantic"].numpy() - g["logits"]).max() <= 1e-6
    assert (g["logits"] < 0).any() and (g["logits"] > 0.3).any()  # a clamp(0, 1) behind it bites, and the output is not flat


def test_rrdnet_restatement_matches_reference():
    """oracle.rrdnet_forward's three stacks against the reference's own RRDNet module (mon_extra/.../rrdnet/model/RRDNet.py)."""
    g = np.load(os.path.join(GOLDEN, "rrdnet_seed71_2x37x53.npz"))
    sd_ref = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    m = mon_amd.MODELS.build(name="rrdnet_re", weights=dict(sd_ref))
    assert set(m.state_dict()) == set(sd_ref)
    x = torch.from_numpy(g["image"])
    out = og.rrdnet_forward(m.state_dict(), x, 0.4)
    for k in ("illumination", "reflectance", "noise"):
        assert np.abs(out[k].numpy() - g[k]).max() <= 1e-6, k
    want = torch.clamp(torch.pow(torch.from_numpy(g["illumination"]), 0.4) * ((x - torch.from_numpy(g["noise"])) / torch.from_numpy(g["illumination"])), 0, 1)
    assert torch.equal(out["enhanced"], want)


def test_hvi_transform_restatement_properties():
    """oracle.rgb_to_hvi / hvi_to_rgb (core/image/color.py:313-434; parity unpinned -- `mon.core` is not importable here): known values and the algebra the
    reference relies on.  Grey pixels sit on the intensity axis, black maps to the origin, the inverse undoes the forward transform on in-gamut colours, the
    intensity channel is max(R, G, B), and the radius (sin(v pi / 2) + eps)^k shrinks the colour plane for dark pixels."""
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 16, 20, generator=g)
    x[0, :, 0, :] = x[0, :1, 0, :]              # greys
    x[0, :, 1, :] = 0.0                         # black
    hvi = og.rgb_to_hvi(x, 0.2)
    assert torch.equal(hvi[:, 2], x.max(1)[0])
    assert float(hvi[0, :2, 0].abs().max()) == 0.0 and float(hvi[0, :, 1].abs().max()) == 0.0
    assert float((og.hvi_to_rgb(hvi, 0.2) - x).abs().max()) <= 2e-6
    pure = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]).t().reshape(1, 3, 1, 3)      # R, G, B at full value: hue 0, 1/3, 2/3, radius 1
    h = og.rgb_to_hvi(pure, 0.2)[0]
    want = torch.tensor([[1.0, -0.5, -0.5], [0.0, 3 ** 0.5 / 2, -(3 ** 0.5) / 2], [1.0, 1.0, 1.0]]).reshape(3, 1, 3)
    assert float((h - want).abs().max()) <= 1e-6
    dark, bright = og.rgb_to_hvi(pure * 0.05, 0.2)[0, :2], og.rgb_to_hvi(pure, 0.2)[0, :2]
    assert float(dark.norm(dim=0).max()) < float(bright.norm(dim=0).min())
    assert float((og.rgb_to_hvi(pure * 0.05, 1.0)[0, :2].norm(dim=0) - (torch.sin(torch.tensor(0.05 * 3.141592653589793 / 2)) + 1e-8)).abs().max()) <= 1e-6


# ---- round 4: goldens produced by the reference's own class / function bodies, taken out of files whose imports cannot be satisfied here (make_golden._extract) ----
def _retinex_golden():
    g = np.load(os.path.join(GOLDEN, "retinexnet_c16_seed81_2x42x58.npz"))
    sd = {k[len("w.decom."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.decom.")}
    sd.update({k[len("w.relight."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.relight.")})
    return g, sd


def test_retinexnet_restatement_matches_reference():
    """oracle.generators.retinex_decom / retinexnet_forward / retinexnet_losses against RetinexNet.forward(input_low, input_high) of
    src/mon_extra/vision/enhance/llie/retinexnet/model.py:21-90, 199-238 run on CPU: both sub-networks, every output it keeps, both losses and all nine terms."""
    g, sd = _retinex_golden()
    low, high = torch.from_numpy(g["low"]), torch.from_numpy(g["high"])
    out = og.retinexnet_forward(sd, low)
    assert np.abs(out["reflectance"].numpy() - g["R_low"]).max() <= 1e-6
    assert np.abs(out["illumination"].numpy() - g["I_low_3"][:, :1]).max() <= 1e-6
    assert np.abs(out["illumination_delta"].numpy() - g["I_delta_3"][:, :1]).max() <= 2e-5 * np.abs(g["I_delta_3"]).max()
    assert np.abs(out["enhanced"].numpy() - g["S"]).max() <= 2e-5 * np.abs(g["S"]).max()
    r_high, i_high = og.retinex_decom(sd, high)
    assert np.abs(r_high.numpy() - g["R_high"]).max() <= 1e-6 and np.abs(i_high.numpy() - g["I_high"]).max() <= 1e-6
    ls = og.retinexnet_losses(sd, low, high)
    assert abs(float(ls["loss_decom"]) - float(g["loss_Decom"])) <= 1e-5 * abs(float(g["loss_Decom"]))
    assert abs(float(ls["loss_relight"]) - float(g["loss_Relight"])) <= 1e-5 * abs(float(g["loss_Relight"]))
    assert abs(float(og.retinex_smooth(out["illumination"], out["reflectance"])) - float(g["Ismooth_loss_low"])) <= 1e-5 * float(g["Ismooth_loss_low"])
    assert abs(float(og.retinex_smooth(out["illumination_delta"], out["reflectance"])) - float(g["Ismooth_loss_delta"])) <= 1e-5 * float(g["Ismooth_loss_delta"])
    # the model object takes the reference's two state dicts under their own key names and reports the same structure
    m = mon_amd.MODELS.build(name="retinexnet", channels=16, weights=dict(sd))
    assert all(torch.equal(m.state_dict()[k], v) for k, v in sd.items())


def test_ssim_restatement_matches_reference():
    """oracle.train.ssim / gauss_1d against custom_ssim / _custom_ssim / _fspecial_gauss_1d of src/mon/nn/metric/image.py:90-237 (what SSIMLoss and BASELINE
    config 5's L1 + SSIM loss evaluate): three image pairs x data_range 1 / 255 x non_negative_ssim, means and per-image values."""
    from oracle import train as otr
    g = np.load(os.path.join(GOLDEN, "ssim_seed91_2x40x56.npz"))
    assert np.abs(otr.gauss_1d(11, 1.5).numpy().ravel() - g["window"].ravel()).max() <= 1e-7
    for name in ("noisy", "dark", "inverse"):
        x, y = torch.from_numpy(g[f"{name}.x"]), torch.from_numpy(g[f"{name}.y"])
        for dr in (1.0, 255.0):
            for nn_ in (False, True):
                want = float(g[f"{name}.dr{int(dr)}.nn{int(nn_)}.mean"])
                got = float(otr.ssim(x, y, data_range=dr, non_negative=nn_))
                assert abs(got - want) <= 2e-6 * max(1.0, abs(want)), (name, dr, nn_, got, want)
                assert abs(float(otr.ssim_loss(x, y, data_range=dr, non_negative=nn_)) - (1.0 - want)) <= 2e-6
                per = g[f"{name}.dr{int(dr)}.nn{int(nn_)}.per_image"]
                for i in range(x.shape[0]):
                    assert abs(float(otr.ssim(x[i:i + 1], y[i:i + 1], data_range=dr, non_negative=nn_)) - float(per[i])) <= 2e-6 * max(1.0, abs(float(per[i])))
        # the classes the loss is built from: CustomSSIM (metric/image.py:316-346) and L1Loss (nn/loss/base.py:164-178), executed out of their files
        assert abs(float(otr.ssim(x, y, data_range=1.0)) - float(g[f"{name}.class.dr1"])) <= 2e-6 and float(g[f"{name}.class.dr1"]) == float(g[f"{name}.dr1.nn0.mean"])
        assert abs(float(otr.ssim(x, y, data_range=255.0, non_negative=True)) - float(g[f"{name}.class.dr255.nn1"])) <= 2e-6
        assert abs(float(otr.l1_loss(x, y)) - float(g[f"{name}.l1.mean"])) <= 1e-7 and abs(float(otr.l1_loss(x, y, 0.5)) - float(g[f"{name}.l1.w05"])) <= 1e-7
        assert abs(float((x - y).abs().sum()) - float(g[f"{name}.l1.sum"])) <= 1e-5 * float(g[f"{name}.l1.sum"])
        want = float(g[f"{name}.l1.mean"]) + 1.0 - float(g[f"{name}.class.dr1"])
        assert abs(float(otr.l1_ssim_loss(x, y, 1.0, 1.0, 1.0)) - want) <= 3e-6


def test_zero_dce_loss_restatement_matches_reference():
    """oracle.train's four zero-reference terms against L_TV / L_spa / L_color / L_exp of the original repository's myloss.py run on the reference network's
    outputs, and their sum as my_train.py:80-85 weighs it.  mon's TotalVariationLoss (nn/loss/image.py:1355-1392, what the oracle restates) counts the channels
    into its denominators: L_TV = C x mon's value."""
    from oracle import train as otr
    from oracle import zero_dce as o_zdce
    g = np.load(os.path.join(GOLDEN, "zdce_losses_seed5_2x64x80.npz"))
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    x = torch.from_numpy(g["image"])
    out = o_zdce.forward(sd, x)
    enh, adj = out["enhanced"], out["adjust"]
    assert abs(float(enh.mean()) - float(g["enhanced_mean"])) <= 1e-6 and abs(float(adj.abs().mean()) - float(g["adjust_abs_mean"])) <= 1e-6
    tv, spa = float(otr.total_variation_loss(adj)) * adj.shape[1], float(otr.spatial_consistency_loss(enh, x))
    col, exp = float(otr.color_constancy_loss(enh)), float(otr.exposure_control_loss(enh, 16, 0.6))
    for got, key in ((tv, "L_TV"), (spa, "L_spa"), (col, "L_color"), (exp, "L_exp")):
        assert abs(got - float(g[key])) <= 2e-5 * abs(float(g[key])) + 1e-10, (key, got, float(g[key]))
    assert abs(200 * tv + spa + 5 * col + 10 * exp - float(g["total"])) <= 2e-5 * float(g["total"])
    # mon's composite (zero_dce.py:65-79: the same weights, its own TV normalisation)
    mon_total = float(otr.zero_dce_loss(x, adj, enh))
    assert abs(mon_total - (float(g["total"]) - 200 * float(g["L_TV"]) * (1 - 1 / adj.shape[1]))) <= 2e-5 * float(g["total"])


def test_upsampler_restatement_matches_reference():
    """oracle.generators.upsampler_forward against Upsampler(default_conv, scale, n_feats, act) of src/mon/nn/modules/encoder/edsr.py:88-123 (scale 2 and 4, with / without ReLU)."""
    g = np.load(os.path.join(GOLDEN, "upsampler_edsr.npz"))
    for tag in ("s2_c16", "s4_c8_relu", "s2_c32_relu"):
        sd = {k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + ".w.")}
        y = og.upsampler_forward(sd, torch.from_numpy(g[f"{tag}.x"]), relu=bool(g[f"{tag}.relu"]))
        assert tuple(y.shape) == tuple(g[f"{tag}.y"].shape)
        assert np.abs(y.numpy() - g[f"{tag}.y"]).max() <= 1e-6, tag


def test_hvi_transform_restatement_matches_reference():
    """oracle.generators.rgb_to_hvi / hvi_to_rgb against RGBToHVI.rgb_to_hvi / hvi_to_rgb of src/mon/core/image/color.py:313-434 (class body executed out of the file:
    make_golden._extract), density_k 0.2 and 0.35, in-gamut and perturbed HVI tensors."""
    g = np.load(os.path.join(GOLDEN, "hvi_transform_seed121.npz"))
    x = torch.from_numpy(g["image"])
    for k, tag in ((0.2, "k20"), (0.35, "k35")):
        assert np.abs(og.rgb_to_hvi(x, k).numpy() - g[f"{tag}.hvi"]).max() <= 1e-6
        assert np.abs(og.hvi_to_rgb(torch.from_numpy(g[f"{tag}.hvi"]), k).numpy() - g[f"{tag}.rgb"]).max() <= 1e-6
        assert np.abs(og.hvi_to_rgb(torch.from_numpy(g[f"{tag}.pert"]), k).numpy() - g[f"{tag}.rgb_pert"]).max() <= 1e-6


def test_learnable_instance_norm_restatement_matches_reference():
    """oracle.generators.learnable_instance_norm against LearnableInstanceNorm2d.forward of src/mon/nn/modules/normalization.py:177-224 (LLLINet's norm leaf; class
    body executed out of the file by make_golden._extract), r inside and outside [0, 1]."""
    g = np.load(os.path.join(GOLDEN, "learnable_instance_norm.npz"))
    for tag in ("c12", "c32"):
        t = {k: torch.from_numpy(g[f"{tag}.{k}"]) for k in ("x", "weight", "bias", "r")}
        y = og.learnable_instance_norm(t["x"], t["weight"], t["bias"], t["r"])
        assert np.abs(y.numpy() - g[f"{tag}.y"]).max() <= 1e-6, tag


def test_mprnet_restatement_matches_reference_blocks():
    """oracle.generators.mprnet_forward against the reference's own MPRNet blocks (conv / CAB / SAM / Encoder with and without CSFF / Decoder / ORSNet of
    src/mon/vision/enhance/multitask/mprnet/mprnet.py:34-357, executed out of the file by make_golden.gen_mprnet and called in the order of MPRNet.forward,
    mprnet.py:483-553): the three outputs and the block-level taps."""
    g = np.load(os.path.join(GOLDEN, "mprnet_c8_seed141_2x24x32.npz"))
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    assert len(sd) == 226 and "stage2_encoder.csff_dec3.weight" in sd and tuple(sd["shallow_feat1.1.body.1.weight"].shape) == (1,)
    taps = {}
    out = og.mprnet_forward(sd, torch.from_numpy(g["image"]), num_cab=2, taps=taps)
    for k in ("stage1", "stage2", "enhanced"):
        assert np.abs(out[k].numpy() - g[k]).max() <= 2e-5 * max(1.0, np.abs(g[k]).max()), k
    for nm in ("feat1_ltop", "res1_top", "feat2_top", "res2"):
        for i in range(3):
            want = g[f"tap.{nm}.{i}"]
            assert np.abs(taps[nm][i].numpy() - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), (nm, i)
    for nm in ("sam12_top_feats", "sam12_top_img", "orsnet"):
        want = g[f"tap.{nm}"]
        assert np.abs(taps[nm].numpy() - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), nm


def test_lllinet_restatement_matches_reference_blocks():
    """oracle.generators.lllinet_forward against the reference's own UNetConvBlock (lllinet.py:97-156, with LearnableInstanceNorm2d and SimAM, all executed out of
    their files by make_golden.gen_lllinet) on the grid of LLLINet.forward (lllinet.py:229-260) at widths 2 .. 32: output before and after the clamp, block-level taps."""
    g = np.load(os.path.join(GOLDEN, "lllinet_w2_seed151_2x32x48.npz"))
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    assert "conv0_0.norm1.r" not in sd and tuple(sd["conv3_1.norm1.r"].shape) == (48,)
    x = torch.from_numpy(g["image"])
    assert np.abs(og.lllinet_forward(sd, x, clamp=False)["enhanced"].numpy() - g["y"]).max() <= 2e-5
    assert np.abs(og.lllinet_forward(sd, x)["enhanced"].numpy() - g["enhanced"]).max() <= 2e-5
    x0_0 = og._lllinet_block(sd, "conv0_0", x)
    assert np.abs(x0_0.numpy() - g["tap.x0_0"]).max() <= 1e-5
    x1_0 = og._lllinet_block(sd, "conv1_0", torch.nn.functional.max_pool2d(x0_0, 2, 2))
    assert np.abs(x1_0.numpy() - g["tap.x1_0"]).max() <= 1e-5


def test_guided_filter_restatement_matches_reference():
    """oracle.filters.box_filter / guided_filter against box_filter / BoxFilter (src/mon/vision/filtering/box_filter.py:27-177) and GuidedFilter
    (guided_filter.py:99-146) executed out of their files: radii 1 / 3 / 5, eps 1e-2 .. 1e-8, three- and one-channel image against a three-channel guide."""
    from oracle import filters as of
    g = np.load(os.path.join(GOLDEN, "guided_filter_seed161.npz"))
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    for r in (1, 3):
        assert np.abs(of.box_filter(x, r).numpy() - g[f"box_r{r}"]).max() <= 1e-5
    assert np.abs(of.box_filter(y, 2).numpy() - g["box_k5"]).max() <= 1e-5
    # The class hands ``radius`` to BoxFilter as its kernel SIZE (guided_filter.py:115, box_filter.py:163-172): GuidedFilter(3) averages 3x3 windows, GuidedFilter(1)
    # returns the guide.  The fixture caught the restatement (and the engine's GCENet) filtering with 7x7 windows for gf_radius = 3.
    for r, eps, tol in ((3, 1e-4, 1e-5), (1, 1e-2, 1e-6), (5, 1e-8, 1e-3)):       # (eps 1e-8: A = cov / (var + eps) amplifies rounding where the window is flat)
        assert np.abs(of.guided_filter_module(x, y, r, eps).numpy() - g[f"gf_r{r}_eps{eps:g}"]).max() <= tol, (r, eps)
    assert np.abs(of.guided_filter_module(x[:, :1], y, 3, 1e-4).numpy() - g["gf_r3_gray_guide3"]).max() <= 1e-5
    assert np.abs(of.guided_filter(x, y, 3, 1e-4).numpy() - g["gf_r3_eps0.0001"]).max() > 1e-2 and np.abs(g["gf_r1_eps0.01"] - g["y"]).max() <= 1e-5


def test_zero_dce_loss_restatement_matches_mon_classes():
    """oracle.train's zero-reference terms and composite against mon's OWN classes: Loss of zero_dce.py:32-79 over ColorConstancyLoss / ExposureControlLoss /
    SpatialConsistencyLoss / TotalVariationLoss (nn/loss/image.py) on nn/loss/base.py's Loss, all executed out of their files (make_golden.gen_mon_zero_dce_loss)."""
    from oracle import train as otr
    from oracle import zero_dce as o_zdce
    g = np.load(os.path.join(GOLDEN, "zdce_loss_mon_seed5_2x64x80.npz"))
    sd = o_zdce.init_state_dict(seed=int(g["seed"]), gain=float(g["gain"]))
    x = torch.from_numpy(g["image"])
    out = o_zdce.forward(sd, x)
    enh, adj = out["enhanced"], out["adjust"]
    rel = lambda got, key: abs(float(got) - float(g[key])) <= 2e-5 * abs(float(g[key])) + 1e-10
    assert rel(otr.spatial_consistency_loss(enh, x), "spa") and rel(otr.exposure_control_loss(enh, 16, 0.6), "exp")
    assert rel(otr.color_constancy_loss(enh), "col") and rel(otr.total_variation_loss(adj), "tva")
    assert list(g["weights"]) == [1.0, 10.0, 5.0, 200.0] and rel(otr.zero_dce_loss(x, adj, enh), "total")
    assert rel(otr.exposure_control_loss(enh, 8, 0.5), "exp_p8_e05") and rel(3.0 * otr.total_variation_loss(adj), "tva_w3")
    assert rel(otr.spatial_consistency_loss(enh, x) * (2 * 16 * 20), "spa_sum")          # reduction "sum": the 16 x 20 pooled map of both frames
    m = enh.mean((2, 3), keepdim=True)
    mr, mg, mb = torch.split(m, 1, dim=1)
    want = torch.pow(((mr - mg) ** 2) ** 2 + ((mr - mb) ** 2) ** 2 + ((mb - mg) ** 2) ** 2, 0.5)
    assert np.abs(want.numpy() - g["col_none"]).max() <= 1e-9
