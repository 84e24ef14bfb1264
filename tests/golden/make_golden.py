#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REFERENCE's own model files.

Runs ONLY in the build container (needs /root/reference).  It importlib-loads
the pure-torch reference files
    src/mon_extra/vision/enhance/llie/zero_dce/model.py      (enhance_net_nopool)
    src/mon_extra/vision/enhance/llie/zero_dce++/model.py    (enhance_net_nopool, CSDN_Tem)
    src/mon_extra/vision/enhance/llie/llunet++/model.py      (NestedUNet)
    src/mon_extra/vision/enhance/multitask/hinet/basicsr/models/archs/hinet_arch.py   (HINet)
    src/mon_extra/vision/enhance/llie/zero_ig/model.py       (Enhancer, Denoise_1, Denoise_2, Finetunemodel.forward)
    src/mon_extra/vision/enhance/llie/mtfe/unet_model.py     (UNet: DoubleConv / Down / Up / OutConv of unet_parts.py)
    src/mon_extra/vision/enhance/llie/rrdnet/model/RRDNet.py (RRDNet)
    src/mon/nn/modules/encoder/edsr.py                       (Upsampler: conv3x3 + nn.PixelShuffle; round 4)
and (round 4) executes, out of files whose module-level imports are not installed here, the pure-torch definitions
    DecomNet / RelightNet / RetinexNet of src/mon_extra/vision/enhance/llie/retinexnet/model.py,
    custom_ssim / _custom_ssim / _gaussian_filter / _fspecial_gauss_1d of src/mon/nn/metric/image.py,
    L_color / L_spa / L_exp / L_TV of src/mon_extra/vision/enhance/llie/zero_dce/myloss.py,
    RGBToHVI of src/mon/core/image/color.py  (see _extract below),
loads seeded weights into them, runs their forward on CPU fp32 and stores
inputs / weights / outputs as small fixtures.  The fixtures are data only; no
reference source is copied.  Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch

REF = os.environ.get("IE_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import zero_dce as o_zdce  # noqa: E402  (seeded weight rule lives in the oracle)


def _load(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _sample_crop(h, w, y0=200, x0=250):
    """Crop of the reference's sample low-light image project/enhance/data/10.png."""
    from PIL import Image
    im = np.asarray(Image.open(os.path.join(REF, "project/enhance/data/10.png")).convert("RGB"))
    crop = im[y0:y0 + h, x0:x0 + w].astype(np.float32) / 255.0  # to_image_tensor(normalize): /255
    return torch.from_numpy(crop).permute(2, 0, 1)[None].contiguous()


def _u8(x):
    return x.mul(255).add(0.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()


def gen_zero_dce():
    mod = _load("src/mon_extra/vision/enhance/llie/zero_dce/model.py", "ref_zero_dce")
    torch.manual_seed(0)
    cases = {
        # name: (state_dict, input)
        "init_rand_1x64x96": (o_zdce.init_state_dict(seed=1), torch.rand(1, 3, 64, 96)),
        "trained_lowlight_2x40x72": (o_zdce.init_state_dict(seed=2, gain=4.0),
                                     torch.rand(2, 3, 40, 72) ** 2.2 * 0.25),
        "trained_sample_1x48x80": (o_zdce.init_state_dict(seed=3, gain=4.0), _sample_crop(48, 80)),
        # ragged: W not a multiple of any tile width, H odd
        "trained_rand_1x37x131": (o_zdce.init_state_dict(seed=4, gain=3.0), torch.rand(1, 3, 37, 131)),
    }
    for name, (sd, x) in cases.items():
        net = mod.enhance_net_nopool()
        net.load_state_dict(sd, strict=True)
        net.eval()
        with torch.no_grad():
            e1, e, r = net(x)
        out = {f"w.{k}": v.numpy() for k, v in sd.items()}
        out.update(image=x.numpy(), enhance_image_1=e1.numpy(), enhanced=e.numpy(), adjust=r.numpy(),
                   enhanced_u8=_u8(e))
        path = os.path.join(HERE, f"zero_dce_{name}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB",
              "| |r|max", float(r.abs().max()), "enh range", float(e.min()), float(e.max()))


from tests.helpers import seeded_state_dict as _seeded  # noqa: E402


def gen_zero_dcepp():
    mod = _load("src/mon_extra/vision/enhance/llie/zero_dce++/model.py", "ref_zero_dcepp")
    for name, scale, x, gain in (("s1_1x40x56", 1, torch.rand(1, 3, 40, 56, generator=torch.Generator().manual_seed(5)), 8.0),
                                 ("s2_2x48x64", 2, torch.rand(2, 3, 48, 64, generator=torch.Generator().manual_seed(6)) ** 2.2 * 0.3, 8.0)):
        net = mod.enhance_net_nopool(scale)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        sd = _seeded(shapes, seed=21 + scale, std=0.02 * gain)
        net.load_state_dict(sd, strict=True)
        net.eval()
        with torch.no_grad():
            e, r = net(x)
        out = {f"w.{k}": v.numpy() for k, v in sd.items()}       # original key names (depth_conv / point_conv)
        out.update(image=x.numpy(), enhanced=e.numpy(), adjust=r.numpy(), scale_factor=np.array(scale))
        path = os.path.join(HERE, f"zero_dcepp_{name}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB | |r|max", float(r.abs().max()))


def gen_llunetpp():
    """LLUnet++ has 29 M parameters: the weights are NOT stored; they are regenerated from the seed by
    oracle-side code (tests/helpers.llunetpp_seeded_weights) and only input/output travel."""
    mod = _load("src/mon_extra/vision/enhance/llie/llunet++/model.py", "ref_llunetpp")
    net = mod.NestedUNet()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = _seeded(shapes, seed=31)
    net.load_state_dict(sd, strict=True)
    net.eval()
    x = torch.rand(1, 3, 32, 48, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        y = net(x)
    path = os.path.join(HERE, "llunetpp_seed31_1x32x48.npz")
    np.savez_compressed(path, image=x.numpy(), enhanced=y.numpy(), seed=np.array(31),
                        key_order=np.array(list(shapes.keys())), key_shapes=np.array([str(s) for s in shapes.values()]))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB | out range", float(y.min()), float(y.max()), "mean", float(y.mean()))


def gen_hinet():
    """HINet at reduced widths (wf = 8 and 16; the published wf = 64 has 88.7 M parameters): weights regenerated from the
    seed by tests/helpers.seeded_state_dict, only input / both stage outputs travel."""
    mod = _load("src/mon_extra/vision/enhance/multitask/hinet/basicsr/models/archs/hinet_arch.py", "ref_hinet_arch")
    for wf, seed, (h, w) in ((8, 41, (32, 48)), (16, 42, (32, 32))):
        net = mod.HINet(wf=wf)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        sd = _seeded(shapes, seed=seed)
        net.load_state_dict(sd, strict=True)
        net.eval()
        x = torch.rand(1, 3, h, w, generator=torch.Generator().manual_seed(seed + 100))
        with torch.no_grad():
            out_1, out_2 = net(x)
        path = os.path.join(HERE, f"hinet_wf{wf}_seed{seed}_1x{h}x{w}.npz")
        np.savez_compressed(path, image=x.numpy(), stage1=out_1.numpy(), enhanced=out_2.numpy(), seed=np.array(seed), wf=np.array(wf),
                            key_order=np.array(list(shapes.keys())), key_shapes=np.array([str(s) for s in shapes.values()]))
        print("wrote", path, os.path.getsize(path) // 1024, "KiB | params", sum(v.numel() for v in sd.values()),
              "| out range", float(out_2.min()), float(out_2.max()))


def gen_zero_ig():
    """Zero-IG: the reference's Enhancer / Denoise_1 / Denoise_2 and the predicting forward (Finetunemodel.forward ==
    mon's ZeroIG.forward predicting branch, zero_ig.py:434-452).  Finetunemodel.__init__ wants a checkpoint file mapped to
    cuda:0, so the object is assembled from the reference's own sub-modules and its own forward() is called on it."""
    zdir = os.path.join(REF, "src/mon_extra/vision/enhance/llie/zero_ig")
    sys.path.insert(0, zdir)
    try:
        mod = _load("src/mon_extra/vision/enhance/llie/zero_ig/model.py", "ref_zero_ig")
    finally:
        sys.path.remove(zdir)
    net = mod.Finetunemodel.__new__(mod.Finetunemodel)
    torch.nn.Module.__init__(net)
    net.enhance, net.denoise_1, net.denoise_2 = mod.Enhancer(layers=3, channels=64), mod.Denoise_1(chan_embed=48), mod.Denoise_2(chan_embed=48)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(51)
    sd = _seeded({k: s for k, s in shapes.items() if len(s) > 0}, seed=51)
    for k in shapes:                                     # BatchNorm entries: a valid, non-trivial eval-mode state
        if k.endswith("running_var"):
            sd[k] = torch.rand(shapes[k], generator=g) + 0.5
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(shapes[k], generator=g) * 0.1
        elif k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(0)
        elif k.endswith(".1.weight") and len(shapes[k]) == 1:
            sd[k] = 1.0 + torch.randn(shapes[k], generator=g) * 0.1
    net.load_state_dict(sd, strict=True)
    sd = {k: v.clone() for k, v in net.state_dict().items()}   # enhance.conv.* and enhance.blocks.i.* alias ONE module: keep what it holds
    net.eval()
    x = torch.rand(2, 3, 40, 56, generator=torch.Generator().manual_seed(52)) ** 2.2 * 0.4
    xe = torch.rand(1, 3, 24, 40, generator=torch.Generator().manual_seed(53))
    with torch.no_grad():
        h2, h3 = net(x)
        s = net.enhance(xe)
        d1 = net.denoise_1(xe)
        d2 = net.denoise_2(torch.cat([xe, xe.flip(1)], 1))
    out = {f"w.{k}": v.numpy() for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    out.update(image=x.numpy(), enhanced=h2.numpy(), denoise=h3.numpy(), enhancer_in=xe.numpy(), enhancer_out=s.numpy(),
               denoise1_out=d1.numpy(), denoise2_out=d2.numpy())
    path = os.path.join(HERE, "zero_ig_seed51_2x40x56.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB | enhanced range", float(h2.min()), float(h2.max()), "mean", float(h2.mean()),
          "| denoise mean", float(h3.mean()))


def gen_unet():
    """U-Net: the reference tree's pure-torch twin of src/mon/vision/segment/unet/unet.py:35-197 (same DoubleConv / Down /
    Up(bilinear align_corners=True | ConvTranspose2d, pad, cat[skip, up]) / OutConv topology; mon's copy does not run as
    shipped, line 51).  Widths 16..256, convs WITH bias, keys ``*.double_conv.{0,1,3,4}``.  Eval-mode BatchNorm with
    non-trivial running statistics; odd frames so the decoder's zero pad is exercised.  Weights are regenerated from the
    seed (tests/helpers.seeded_bn_state_dict); input / output / one intermediate travel."""
    from tests.helpers import seeded_bn_state_dict
    udir = os.path.join(REF, "src/mon_extra/vision/enhance/llie/mtfe")
    sys.path.insert(0, udir)
    try:
        mod = _load("src/mon_extra/vision/enhance/llie/mtfe/unet_model.py", "ref_mtfe_unet")
    finally:
        sys.path.remove(udir)
    for tag, bilinear, seed, (n, h, w) in (("bilinear", True, 61, (2, 33, 47)), ("convt", False, 62, (1, 35, 50))):
        net = mod.UNet(3, 3, bilinear=bilinear)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        sd = seeded_bn_state_dict(shapes, seed)
        net.load_state_dict(sd, strict=True)
        net.eval()
        x = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(seed + 100))
        with torch.no_grad():
            y = net(x)
            x1 = net.inc(x)
            x5 = net.down4(net.down3(net.down2(net.down1(x1))))
        path = os.path.join(HERE, f"unet_mtfe_{tag}_seed{seed}_{n}x{h}x{w}.npz")
        np.savez_compressed(path, image=x.numpy(), logits=y.numpy(), inc=x1.numpy(), bottleneck=x5.numpy(), seed=np.array(seed),
                            bilinear=np.array(bilinear), key_order=np.array(list(shapes.keys())),
                            key_shapes=np.array([str(s) for s in shapes.values()]))
        print("wrote", path, os.path.getsize(path) // 1024, "KiB | params", sum(v.numel() for k, v in sd.items() if v.dim() > 0),
              "| logits range", float(y.min()), float(y.max()), "in [0,1]:", float(((y > 0) & (y < 1)).float().mean()))


def gen_rrdnet():
    """RRDNet: the original repository's module (three 5-layer stacks, sigmoid / sigmoid / tanh) == mon's RRDNet_RE stacks
    (rrdnet.py:208-240, same Sequential indices) -- the Retinex recombination of rrdnet.py:271-279 is applied to ITS outputs
    with torch ops here (pow, sub, div, mul, clamp), so ``enhanced`` is pinned through the reference's own three maps."""
    mod = _load("src/mon_extra/vision/enhance/llie/rrdnet/model/RRDNet.py", "ref_rrdnet")
    net = mod.RRDNet()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = _seeded(shapes, seed=71)
    net.load_state_dict(sd, strict=True)
    net.eval()
    x = torch.rand(2, 3, 37, 53, generator=torch.Generator().manual_seed(72)) * 0.6 + 0.05
    with torch.no_grad():
        illumination, reflectance, noise = net(x)
    out = {f"w.{k}": v.numpy() for k, v in sd.items()}
    out.update(image=x.numpy(), illumination=illumination.numpy(), reflectance=reflectance.numpy(), noise=noise.numpy())
    path = os.path.join(HERE, "rrdnet_seed71_2x37x53.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB | params", sum(v.numel() for v in sd.values()),
          "| illumination", float(illumination.min()), float(illumination.max()), "noise |max|", float(noise.abs().max()))


# ---- reference files whose module-level imports cannot be satisfied here (cv2, thop, torchmetrics, torchvision, the mon package) ---------------
# Their pure-torch definitions are taken out of the file by NAME through its syntax tree and executed UNMODIFIED (same code object the file would
# give, compiled under the file's own name) with only torch / nn / F (/ numpy / Variable) in scope: the arithmetic that produces the fixture is the
# reference's own, nothing of it is restated or stored.  `.cuda()` -- hard-coded in some forward() bodies -- is the identity while they run.
def _extract(rel, names, ns):
    import ast
    path = os.path.join(REF, rel)
    tree = ast.parse(open(path).read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert sorted(n.name for n in keep) == sorted(names), (rel, [n.name for n in keep])
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


class _no_cuda:
    def __enter__(self):
        self.saved = (torch.Tensor.cuda, torch.nn.Module.cuda)
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda, torch.nn.Module.cuda = self.saved
        return False


def gen_retinexnet():
    """RetinexNet (BASELINE config 4): DecomNet / RelightNet / RetinexNet.forward with its two losses, out of
    src/mon_extra/vision/enhance/llie/retinexnet/model.py (the file imports cv2, thop and mon).  The two sub-networks are built at channel = 16
    (their own constructor argument; 64 is 2.2 MB of weights) and put into a RetinexNet() instance, whose forward(input_low, input_high) then runs
    as shipped: DecomNet on both frames, RelightNet(I_low, R_low), every loss term, output_S = R_low * I_delta_3."""
    import torch.nn as nn
    import torch.nn.functional as F
    from torch.autograd import Variable
    ns = _extract("src/mon_extra/vision/enhance/llie/retinexnet/model.py", ["DecomNet", "RelightNet", "RetinexNet"],
                  {"torch": torch, "nn": nn, "F": F, "np": np, "Variable": Variable})
    torch.manual_seed(81)
    with _no_cuda():
        net = ns["RetinexNet"]()
        net.DecomNet, net.RelightNet = ns["DecomNet"](channel=16), ns["RelightNet"](channel=16)
        with torch.no_grad():       # "trained-like" weights: torch's default init x 3 on the conv kernels, so R / L / I_delta vary over the frame instead of sitting at sigmoid(0)
            for m_ in list(net.DecomNet.modules()) + list(net.RelightNet.modules()):
                if isinstance(m_, nn.Conv2d):
                    m_.weight.mul_(3.0)
        g = torch.Generator().manual_seed(82)
        high = torch.rand(2, 3, 42, 58, generator=g)
        low = (high ** 2.2 * 0.3 + 0.02 * torch.rand(2, 3, 42, 58, generator=g)).clamp(0, 1)
        net.forward(low.numpy(), high.numpy())          # (no eval(): the class overrides nn.Module.train with its training loop; it has no mode-dependent layer)
        r_high, i_high = net.DecomNet(high)
    out = {f"w.decom.{k}": v.detach().numpy() for k, v in net.DecomNet.state_dict().items()}
    out.update({f"w.relight.{k}": v.detach().numpy() for k, v in net.RelightNet.state_dict().items()})
    out.update(low=low.numpy(), high=high.numpy(), R_low=net.output_R_low.numpy(), I_low_3=net.output_I_low.numpy(), I_delta_3=net.output_I_delta.numpy(),
               S=net.output_S.numpy(), R_high=r_high.detach().numpy(), I_high=i_high.detach().numpy(),
               loss_Decom=np.array(float(net.loss_Decom)), loss_Relight=np.array(float(net.loss_Relight)),
               **{k: np.array(float(getattr(net, k))) for k in ("recon_loss_low", "recon_loss_high", "recon_loss_mutal_low", "recon_loss_mutal_high", "equal_R_loss",
                                                                 "relight_loss", "Ismooth_loss_low", "Ismooth_loss_high", "Ismooth_loss_delta")})
    path = os.path.join(HERE, "retinexnet_c16_seed81_2x42x58.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB | loss_Decom", float(net.loss_Decom), "loss_Relight", float(net.loss_Relight), "S range", float(net.output_S.min()), float(net.output_S.max()))


def gen_ssim():
    """custom_ssim / _custom_ssim / _gaussian_filter / _fspecial_gauss_1d of src/mon/nn/metric/image.py:90-237 (the module imports torchmetrics and mon.globals):
    what SSIMLoss (nn/loss/image.py:834-866) and BASELINE config 5's L1 + SSIM loss compute."""
    import torch.nn.functional as F
    ns = _extract("src/mon/nn/metric/image.py", ["_fspecial_gauss_1d", "_gaussian_filter", "_custom_ssim", "custom_ssim"], {"torch": torch, "F": F})
    g = torch.Generator().manual_seed(91)
    a = torch.rand(2, 3, 40, 56, generator=g)
    pairs = {"noisy": (a, (a + 0.1 * torch.randn(2, 3, 40, 56, generator=g)).clamp(0, 1)), "dark": (a ** 2.2 * 0.4, a), "inverse": (a, 1.0 - a)}
    out = {}
    for name, (x, y) in pairs.items():
        out[f"{name}.x"], out[f"{name}.y"] = x.numpy(), y.numpy()
        for dr in (1.0, 255.0):
            for nn_ in (False, True):
                tag = f"{name}.dr{int(dr)}.nn{int(nn_)}"
                out[tag + ".mean"] = np.array(float(ns["custom_ssim"](x, y, data_range=dr, size_average=True, non_negative_ssim=nn_)))
                out[tag + ".per_image"] = ns["custom_ssim"](x, y, data_range=dr, size_average=False, non_negative_ssim=nn_).numpy()
        win = ns["_fspecial_gauss_1d"](11, 1.5).repeat([3, 1, 1, 1])
        spc, cs = ns["_custom_ssim"](x, y, data_range=1.0, window=win, size_average=False)
        out[f"{name}.ssim_per_channel"], out[f"{name}.cs_per_channel"] = spc.numpy(), cs.numpy()
    out["window"] = ns["_fspecial_gauss_1d"](11, 1.5).numpy()
    # the two CLASSES BASELINE config 5's loss is made of: CustomSSIM (metric/image.py:316-346, what SSIMLoss.forward calls: loss = 1 - ssim) and L1Loss on its base class
    # (nn/loss/base.py:61-117, 164-178; registered through the reference's Factory, core/factory.py:26-133)
    import copy, inspect, types
    from abc import ABC, abstractmethod
    from typing import Any, Literal
    import torch.nn as nn
    from torch.nn.modules.loss import _Loss
    ns["nn"] = nn
    cls = _extract("src/mon/nn/metric/image.py", ["CustomSSIM"], ns)["CustomSSIM"]
    Factory = _extract("src/mon/core/factory.py", ["Factory"], {"copy": copy, "inspect": inspect, "Any": Any})["Factory"]
    b = _extract("src/mon/nn/loss/base.py", ["reduce_loss", "Loss", "L1Loss"], {"torch": torch, "nn": nn, "F": F, "_Loss": _Loss, "ABC": ABC, "abstractmethod": abstractmethod,
                                                                                "Literal": Literal, "LOSSES": Factory("Losses")})
    for name, (x, y) in pairs.items():
        out[f"{name}.class.dr1"] = np.array(float(cls(data_range=1.0)(x, y)))
        out[f"{name}.class.dr255.nn1"] = np.array(float(cls(non_negative_ssim=True)(x, y)))
        out[f"{name}.l1.mean"], out[f"{name}.l1.sum"] = np.array(float(b["L1Loss"]()(x, y))), np.array(float(b["L1Loss"](reduction="sum")(x, y)))
        out[f"{name}.l1.w05"] = np.array(float(b["L1Loss"](loss_weight=0.5)(x, y)))
    path = os.path.join(HERE, "ssim_seed91_2x40x56.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB |", {k: float(v) for k, v in out.items() if k.endswith("dr1.nn0.mean")})


def gen_zero_dce_losses():
    """The four zero-reference loss terms of the original Zero-DCE repository (L_color, L_spa, L_exp, L_TV of src/mon_extra/vision/enhance/llie/zero_dce/myloss.py; the module
    imports torchvision) on the outputs of the reference network, combined as its trainer does (my_train.py:80-85: 200 L_TV(A) + mean L_spa + 5 mean L_color + 10 mean L_exp).
    mon's own classes (nn/loss/image.py:191-217, 534-568, 904-1270, 1355-1392) are these with ONE difference the fixture records: TotalVariationLoss counts the
    channels into its denominators (L_TV / C)."""
    import torch.nn as nn
    import torch.nn.functional as F
    ns = _extract("src/mon_extra/vision/enhance/llie/zero_dce/myloss.py", ["L_color", "L_spa", "L_exp", "L_TV"], {"torch": torch, "nn": nn, "F": F})
    mod = _load("src/mon_extra/vision/enhance/llie/zero_dce/model.py", "ref_zero_dce_for_losses")
    sd = o_zdce.init_state_dict(seed=5, gain=3.0)
    net = mod.enhance_net_nopool()
    net.load_state_dict(sd, strict=True)
    net.eval()
    x = torch.rand(2, 3, 64, 80, generator=torch.Generator().manual_seed(101)) ** 2.2 * 0.5
    with torch.no_grad(), _no_cuda():
        _, enhanced, A = net(x)
        l_color, l_spa, l_exp, l_tv = ns["L_color"](), ns["L_spa"](), ns["L_exp"](16, 0.6), ns["L_TV"]()
        tv, spa, col, exp = l_tv(A), torch.mean(l_spa(enhanced, x)), torch.mean(l_color(enhanced)), torch.mean(l_exp(enhanced))
        total = 200 * tv + spa + 5 * col + 10 * exp
    out = {f"w.{k}": v.numpy() for k, v in sd.items()}
    # (enhanced / adjust are not stored: the network's forward is pinned by zero_dce_*.npz and the tests recompute them from the weights)
    out.update(image=x.numpy(), L_TV=np.array(float(tv)), L_spa=np.array(float(spa)), L_color=np.array(float(col)),
               L_exp=np.array(float(exp)), total=np.array(float(total)), enhanced_mean=np.array(float(enhanced.mean())), adjust_abs_mean=np.array(float(A.abs().mean())))
    with _no_cuda():
        out["L_spa_map"] = ns["L_spa"]()(enhanced, x).numpy()
    path = os.path.join(HERE, "zdce_losses_seed5_2x64x80.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB | tv", float(tv), "spa", float(spa), "col", float(col), "exp", float(exp), "total", float(total))


def gen_hvi():
    """RGBToHVI.rgb_to_hvi / hvi_to_rgb of src/mon/core/image/color.py:313-434 (the colour transform around LLLINet's grid in `lllinet_hvi`; the module imports cv2 and
    kornia, the class is pure torch): a random frame with grey, black and saturated rows, density_k = 0.2 (the parameter's initial value) and 0.35."""
    import torch.nn as nn
    ns = _extract("src/mon/core/image/color.py", ["RGBToHVI"], {"torch": torch, "nn": nn})
    g = torch.Generator().manual_seed(121)
    x = torch.rand(2, 3, 18, 24, generator=g)
    x[0, :, 0, :] = x[0, :1, 0, :]                   # greys
    x[0, :, 1, :] = 0.0                              # black
    x[0, 1:, 2, :] = 0.0                             # pure reds of varying value
    x[1] = x[1] ** 2.2 * 0.3                         # a low-light frame
    out = {"image": x.numpy()}
    for k in (0.2, 0.35):
        t = ns["RGBToHVI"]()
        with torch.no_grad():
            t.density_k.fill_(k)
            hvi = t.rgb_to_hvi(x)
            back = t.hvi_to_rgb(hvi)
            # an arbitrary (out-of-gamut) HVI tensor through the inverse: what the network's output looks like before the transform back
            pert = hvi + 0.1 * torch.randn(hvi.shape, generator=torch.Generator().manual_seed(122))
            back_p = t.hvi_to_rgb(pert)
        tag = f"k{int(round(k * 100))}"
        out[f"{tag}.hvi"], out[f"{tag}.rgb"], out[f"{tag}.pert"], out[f"{tag}.rgb_pert"] = hvi.numpy(), back.numpy(), pert.numpy(), back_p.numpy()
    path = os.path.join(HERE, "hvi_transform_seed121.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB | round trip", float((out["k20.rgb"] - x.numpy()).__abs__().max()))


def gen_learnable_instance_norm():
    """LearnableInstanceNorm2d.forward of src/mon/nn/modules/normalization.py:177-224 (LLLINet's norm leaf: y = IN(x)*r + x*(1 - r) with a learnable r per channel; the
    module imports mon, the class is pure torch): random affine parameters and r in [0, 1] and outside it, 12 and 32 channels, an odd frame size."""
    import math
    from typing import Any
    import torch.nn as nn
    import torch.nn.functional as F
    ns = _extract("src/mon/nn/modules/normalization.py", ["LearnableInstanceNorm2d"], {"torch": torch, "nn": nn, "F": F, "math": math, "Any": Any})
    out = {}
    for tag, c, h, w, seed in (("c12", 12, 19, 23, 131), ("c32", 32, 16, 40, 132)):
        g = torch.Generator().manual_seed(seed)
        m = ns["LearnableInstanceNorm2d"](c)
        with torch.no_grad():
            m.weight.copy_(torch.rand(c, generator=g) * 1.5 + 0.25)
            m.bias.copy_(torch.randn(c, generator=g) * 0.3)
            m.r.copy_(torch.rand(c, generator=g) * 1.4 - 0.2)
            x = torch.randn(2, c, h, w, generator=g) * (torch.rand(1, c, 1, 1, generator=g) * 2 + 0.1) + torch.randn(1, c, 1, 1, generator=g)
            y = m(x)
        out[f"{tag}.x"], out[f"{tag}.y"] = x.numpy(), y.numpy()
        out[f"{tag}.weight"], out[f"{tag}.bias"], out[f"{tag}.r"] = m.weight.detach().numpy(), m.bias.detach().numpy(), m.r.detach().numpy()
    path = os.path.join(HERE, "learnable_instance_norm.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_mprnet():
    """MPRNet's building blocks (rank-3 widening): conv / DownSample / UpSample / SkipUpSample / CALayer / CAB / SAM / Encoder / Decoder / ORB / ORSNet of
    src/mon/vision/enhance/multitask/mprnet/mprnet.py:34-357 are plain torch.nn classes inside a module that imports mon; they are executed out of the file.  The model
    class itself derives from mon's Lightning base (and evaluates a mon loss as a default argument), so it cannot be defined here: the blocks are instantiated under the
    attribute names and with the arguments of MPRNet.__init__ (mprnet.py:427-451) and wired as MPRNet.forward does (mprnet.py:483-553) -- that wiring (crops, cats, the
    order of the calls) is a transcription, every arithmetic step runs inside the reference's own block code.  Reduced width (8 / 4 / 4 channels, 2 CABs per ORB)."""
    from typing import Any, Sequence
    import torch.nn as nn
    from torch.nn.common_types import _size_2_t
    ns = _extract("src/mon/vision/enhance/multitask/mprnet/mprnet.py", ["conv", "DownSample", "UpSample", "SkipUpSample", "CALayer", "CAB", "SAM", "Encoder", "Decoder", "ORB", "ORSNet"],
                  {"torch": torch, "nn": nn, "Any": Any, "Sequence": Sequence, "_size_2_t": _size_2_t})
    conv, CAB, SAM, Encoder, Decoder, ORSNet = (ns[k] for k in ("conv", "CAB", "SAM", "Encoder", "Decoder", "ORSNet"))
    cin, c, su, so, ncab, ks, red, bias = 3, 8, 4, 4, 2, 3, 4, False
    torch.manual_seed(141)
    m = nn.Module()                                   # a bag of the reference's blocks under MPRNet's attribute names (= its state_dict keys)
    act = nn.PReLU()
    for k in (1, 2, 3):
        setattr(m, f"shallow_feat{k}", nn.Sequential(conv(cin, c, ks, bias=bias), CAB(c, ks, red, bias=bias, act_layer=act)))
    m.stage1_encoder = Encoder(c, ks, red, act, bias, su, csff=False)
    m.stage1_decoder = Decoder(c, ks, red, act, bias, su)
    m.stage2_encoder = Encoder(c, ks, red, act, bias, su, csff=True)
    m.stage2_decoder = Decoder(c, ks, red, act, bias, su)
    m.stage3_orsnet = ORSNet(c, so, ks, red, act, bias, su, ncab)
    m.sam12, m.sam23 = SAM(c, kernel_size=1, bias=bias), SAM(c, kernel_size=1, bias=bias)
    m.concat12 = conv(c * 2, c, ks, bias=bias)
    m.concat23 = conv(c * 2, c + so, ks, bias=bias)
    m.tail = conv(c + so, cin, ks, bias=bias)
    with torch.no_grad():                             # "trained-like": livelier kernels and gates, the shared PReLU slope away from its default
        for name, p_ in m.named_parameters():
            if name.endswith("body.1.weight"):
                p_.fill_(0.17)
            elif p_.dim() == 4:
                p_.mul_(3.0 if (name.endswith("conv_du.2.weight") or name.endswith("conv3.weight")) else 1.3)
    g = torch.Generator().manual_seed(142)
    x = torch.rand(2, 3, 24, 32, generator=g)
    x[1] = x[1] ** 2.2 * 0.4
    taps = {}
    with torch.no_grad():
        h, w = x.size(2), x.size(3)
        x2top_img, x2bot_img = x[:, :, 0:int(h / 2), :], x[:, :, int(h / 2):h, :]
        x1ltop_img, x1rtop_img = x2top_img[:, :, :, 0:int(w / 2)], x2top_img[:, :, :, int(w / 2):w]
        x1lbot_img, x1rbot_img = x2bot_img[:, :, :, 0:int(w / 2)], x2bot_img[:, :, :, int(w / 2):w]
        feat1_ltop, feat1_rtop = m.stage1_encoder(m.shallow_feat1(x1ltop_img)), m.stage1_encoder(m.shallow_feat1(x1rtop_img))
        feat1_lbot, feat1_rbot = m.stage1_encoder(m.shallow_feat1(x1lbot_img)), m.stage1_encoder(m.shallow_feat1(x1rbot_img))
        feat1_top = [torch.cat((k, v), 3) for k, v in zip(feat1_ltop, feat1_rtop)]
        feat1_bot = [torch.cat((k, v), 3) for k, v in zip(feat1_lbot, feat1_rbot)]
        res1_top, res1_bot = m.stage1_decoder(feat1_top), m.stage1_decoder(feat1_bot)
        x2top_samfeats, stage1_img_top = m.sam12(res1_top[0], x2top_img)
        x2bot_samfeats, stage1_img_bot = m.sam12(res1_bot[0], x2bot_img)
        stage1_img = torch.cat([stage1_img_top, stage1_img_bot], 2)
        x2top_cat = m.concat12(torch.cat([m.shallow_feat2(x2top_img), x2top_samfeats], 1))
        x2bot_cat = m.concat12(torch.cat([m.shallow_feat2(x2bot_img), x2bot_samfeats], 1))
        feat2_top, feat2_bot = m.stage2_encoder(x2top_cat, feat1_top, res1_top), m.stage2_encoder(x2bot_cat, feat1_bot, res1_bot)
        feat2 = [torch.cat((k, v), 2) for k, v in zip(feat2_top, feat2_bot)]
        res2 = m.stage2_decoder(feat2)
        x3_samfeats, stage2_img = m.sam23(res2[0], x)
        x3_cat = m.concat23(torch.cat([m.shallow_feat3(x), x3_samfeats], 1))
        x3_ors = m.stage3_orsnet(x3_cat, feat2, res2)
        stage3_img = m.tail(x3_ors)
        for nm, lst in (("feat1_ltop", feat1_ltop), ("res1_top", res1_top), ("feat2_top", feat2_top), ("res2", res2)):     # block-level taps: Encoder (plain and CSFF), Decoder
            for i, t_ in enumerate(lst):
                taps[f"tap.{nm}.{i}"] = t_.numpy()
        taps["tap.sam12_top_feats"], taps["tap.sam12_top_img"], taps["tap.orsnet"] = x2top_samfeats.numpy(), stage1_img_top.numpy(), x3_ors.numpy()
    out = {f"w.{k}": v.detach().numpy() for k, v in m.state_dict().items()}
    out.update(taps)
    out.update(image=x.numpy(), stage1=stage1_img.numpy(), stage2=stage2_img.numpy(), enhanced=(stage3_img + x).numpy())
    path = os.path.join(HERE, "mprnet_c8_seed141_2x24x32.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB |", len(m.state_dict()), "tensors | enhanced - image", float((stage3_img).abs().max()), "stage1 - stage2",
          float((stage1_img - stage2_img).abs().max()))


def gen_lllinet():
    """LLLINet's block (rank-3 widening): UNetConvBlock of src/mon/vision/enhance/llie/lllinet/lllinet.py:97-156 with the two mon layers it is made of,
    LearnableInstanceNorm2d (nn/modules/normalization.py:177-224) and SimAM (nn/modules/attention.py:1154-1181; its ``act.Sigmoid`` is torch's own class: mon's
    activation module star-imports torch.nn.modules.activation) -- all three executed out of their files.  The block resolves its layers through ``nn.``: the namespace
    it runs in is torch.nn's public names plus those two extracted classes, which is what ``mon.nn`` is for this block.  The model class derives from mon's Lightning base,
    so the fifteen blocks are instantiated under the attribute names of LLLINet.__init__ (lllinet.py:196-219) and called in the order of its forward (:229-260): a
    transcription, at widths 2 .. 32 instead of 32 .. 512 (the class hard-codes them; 32 .. 512 is 120 MB of weights)."""
    import math, types
    from typing import Any
    import torch.nn as nn
    import torch.nn.functional as F
    lin = _extract("src/mon/nn/modules/normalization.py", ["LearnableInstanceNorm2d"], {"torch": torch, "nn": nn, "F": F, "math": math, "Any": Any})["LearnableInstanceNorm2d"]
    simam = _extract("src/mon/nn/modules/attention.py", ["SimAM"], {"torch": torch, "nn": nn, "act": torch.nn.modules.activation})["SimAM"]
    mon_nn = types.SimpleNamespace(**{k: getattr(nn, k) for k in dir(nn) if not k.startswith("_")}, LearnableInstanceNorm2d=lin, SimAM=simam)
    Block = _extract("src/mon/vision/enhance/llie/lllinet/lllinet.py", ["UNetConvBlock"], {"torch": torch, "nn": mon_nn})["UNetConvBlock"]
    f = [2, 4, 8, 16, 32]
    torch.manual_seed(151)
    m = nn.Module()
    m.conv0_0 = Block(3, f[0], use_in=False)
    m.conv1_0, m.conv2_0, m.conv3_0, m.conv4_0 = Block(f[0], f[1]), Block(f[1], f[2]), Block(f[2], f[3]), Block(f[3], f[4])
    m.conv0_1, m.conv1_1, m.conv2_1, m.conv3_1 = Block(f[0] + 2 * f[1], f[0]), Block(f[1] + 2 * f[2], f[1]), Block(f[2] + 2 * f[3], f[2]), Block(f[3] + f[4], f[3])
    m.conv0_2, m.conv1_2, m.conv2_2 = Block(f[0] * 2 + 2 * f[1], f[0]), Block(f[1] * 2 + 2 * f[2], f[1]), Block(f[2] * 2 + f[3], f[2])
    m.conv0_3, m.conv1_3 = Block(f[0] * 3 + 2 * f[1], f[0]), Block(f[1] * 3 + f[2], f[1])
    m.conv0_4 = Block(f[0] * 4 + f[1], f[0])
    m.final = nn.Conv2d(f[0], 3, kernel_size=1)
    g = torch.Generator().manual_seed(152)
    with torch.no_grad():                             # "trained-like": the learnable normalisation away from its initial r = 0.5 / weight 1 / bias 0
        for name, p_ in m.named_parameters():
            if name.endswith("norm1.r"):
                p_.copy_(torch.rand(p_.shape, generator=g) * 0.9 + 0.05)
            elif name.endswith("norm1.weight"):
                p_.copy_(torch.rand(p_.shape, generator=g) + 0.5)
            elif name.endswith("norm1.bias"):
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.2)
            elif p_.dim() == 4:
                p_.mul_(1.4)
        m.final.bias.add_(0.35)
    x = torch.rand(2, 3, 32, 48, generator=g)
    x[1] = x[1] ** 2.2 * 0.3
    pool, up = nn.MaxPool2d(2, 2), nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
    with torch.no_grad():
        x0_0 = m.conv0_0(x)
        x1_0 = m.conv1_0(pool(x0_0)); x2_0 = m.conv2_0(pool(x1_0)); x3_0 = m.conv3_0(pool(x2_0)); x4_0 = m.conv4_0(pool(x3_0))
        x3_1 = m.conv3_1(torch.cat([x3_0, up(x4_0)], 1))
        x2_1 = m.conv2_1(torch.cat([x2_0, up(x3_0), up(x3_1)], 1))
        x1_1 = m.conv1_1(torch.cat([x1_0, up(x2_0), up(x2_1)], 1))
        x0_1 = m.conv0_1(torch.cat([x0_0, up(x1_0), up(x1_1)], 1))
        x2_2 = m.conv2_2(torch.cat([x2_0, x2_1, up(x3_1)], 1))
        x1_2 = m.conv1_2(torch.cat([x1_0, x1_1, up(x2_1), up(x2_2)], 1))
        x0_2 = m.conv0_2(torch.cat([x0_0, x0_1, up(x1_1), up(x1_2)], 1))
        x1_3 = m.conv1_3(torch.cat([x1_0, x1_1, x1_2, up(x2_2)], 1))
        x0_3 = m.conv0_3(torch.cat([x0_0, x0_1, x0_2, up(x1_2), up(x1_3)], 1))
        x0_4 = m.conv0_4(torch.cat([x0_0, x0_1, x0_2, x0_3, up(x1_3)], 1))
        y = m.final(x0_4)
    out = {f"w.{k}": v.detach().numpy() for k, v in m.state_dict().items()}
    out.update(image=x.numpy(), y=y.numpy(), enhanced=torch.clamp(y, 0, 1).numpy(), **{f"tap.{k}": v.numpy() for k, v in (("x0_0", x0_0), ("x1_0", x1_0), ("x3_1", x3_1), ("x0_4", x0_4))})
    path = os.path.join(HERE, "lllinet_w2_seed151_2x32x48.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB |", len(m.state_dict()), "tensors | y range", float(y.min()), float(y.max()), "clamped share",
          float(((y < 0) | (y > 1)).float().mean()))


def gen_guided_filter():
    """The box filter and the guided filter behind GCENet's last step (rank-4 widening): diff_x / diff_y / box_filter / BoxFilter of
    src/mon/vision/filtering/box_filter.py:27-177 and class GuidedFilter of guided_filter.py:99-146 (the modules import cv2, plum and mon; these definitions are pure
    torch) executed out of their files.  FastGuidedFilter's two forwards are plum-dispatched overloads and do not define without plum: not in the fixture."""
    import torch.nn as nn
    from torch.autograd import Variable
    ns = _extract("src/mon/vision/filtering/box_filter.py", ["diff_x", "diff_y", "box_filter", "BoxFilter"], {"torch": torch, "nn": nn, "np": np})
    gf = _extract("src/mon/vision/filtering/guided_filter.py", ["GuidedFilter"], {"torch": torch, "nn": nn, "Variable": Variable, "BoxFilter": ns["BoxFilter"]})["GuidedFilter"]
    g = torch.Generator().manual_seed(161)
    x = torch.rand(2, 3, 26, 34, generator=g)
    x[1] = x[1] ** 2.2 * 0.3
    y = (x * 1.7 + 0.05 * torch.randn(x.shape, generator=g)).clamp(0, 1.5)          # an "enhanced" frame: brighter, noisy
    out = {"x": x.numpy(), "y": y.numpy()}
    with torch.no_grad():
        for r in (1, 3):
            out[f"box_r{r}"] = ns["box_filter"](x, radius=r).numpy()
        out["box_k5"] = ns["BoxFilter"](kernel_size=5)(y).numpy()
        for r, eps in ((3, 1e-4), (1, 1e-2), (5, 1e-8)):
            out[f"gf_r{r}_eps{eps:g}"] = gf(r, eps)(x, y).numpy()
        out["gf_r3_gray_guide3"] = gf(3, 1e-4)(x[:, :1], y).numpy()                # a one-channel image against a three-channel guide (c_x == 1 branch)
    path = os.path.join(HERE, "guided_filter_seed161.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB |", sorted(k for k in out if k not in ("x", "y")))


def gen_mon_zero_dce_loss():
    """mon's OWN zero-reference loss: class Loss of src/mon/vision/enhance/llie/zero_dce/zero_dce.py:32-79 with the four classes it is made of -- ColorConstancyLoss,
    ExposureControlLoss, SpatialConsistencyLoss, TotalVariationLoss (src/mon/nn/loss/image.py:191-217, 534-568, 904-1270, 1355-1392) -- on their base class and
    reduce_loss (nn/loss/base.py:61-117), registered in the reference's own registry class (core/factory.py:26-133, ``LOSSES = Factory("Losses")`` as mon/globals.py builds
    it; with explicit names it never reaches its humps calls).  Every definition is executed out of its file; ``nn.`` resolves to torch.nn's public names plus these
    classes, ``base.`` to the two definitions of loss/base.py.  Inputs: the reference network's outputs on the frame of zdce_losses_seed5 (gen_zero_dce_losses)."""
    import copy, inspect, types
    from abc import ABC, abstractmethod
    from typing import Any, Literal
    import torch.nn as nn
    import torch.nn.functional as F
    from torch.nn.common_types import _size_2_t
    from torch.nn.modules.loss import _Loss
    Factory = _extract("src/mon/core/factory.py", ["Factory"], {"copy": copy, "inspect": inspect, "Any": Any})["Factory"]
    LOSSES = Factory("Losses")
    b = _extract("src/mon/nn/loss/base.py", ["reduce_loss", "Loss"], {"torch": torch, "nn": nn, "F": F, "_Loss": _Loss, "ABC": ABC, "abstractmethod": abstractmethod, "Literal": Literal})
    base = types.SimpleNamespace(reduce_loss=b["reduce_loss"], Loss=b["Loss"])
    names = ["ColorConstancyLoss", "ExposureControlLoss", "SpatialConsistencyLoss", "TotalVariationLoss"]
    li = _extract("src/mon/nn/loss/image.py", names, {"torch": torch, "nn": nn, "F": F, "Literal": Literal, "_size_2_t": _size_2_t, "base": base, "LOSSES": LOSSES})
    assert sorted(LOSSES) == ["color_constancy_loss", "exposure_control_loss", "spatial_consistency_loss", "total_variation_loss", "tv_loss"]
    mon_nn = types.SimpleNamespace(**{k: getattr(nn, k) for k in dir(nn) if not k.startswith("_")}, Loss=base.Loss, **{k: li[k] for k in names})
    ZLoss = _extract("src/mon/vision/enhance/llie/zero_dce/zero_dce.py", ["Loss"], {"torch": torch, "nn": mon_nn, "Literal": Literal})["Loss"]
    mod = _load("src/mon_extra/vision/enhance/llie/zero_dce/model.py", "ref_zero_dce_for_mon_losses")
    sd = o_zdce.init_state_dict(seed=5, gain=3.0)
    net = mod.enhance_net_nopool()
    net.load_state_dict(sd, strict=True)
    net.eval()
    x = torch.rand(2, 3, 64, 80, generator=torch.Generator().manual_seed(101)) ** 2.2 * 0.5       # (the frame of zdce_losses_seed5)
    out = {"image": x.numpy(), "seed": np.array(5), "gain": np.array(3.0)}
    with torch.no_grad():
        _, enhanced, A = net(x)
        loss = ZLoss()
        out["total"] = np.array(float(loss(input=x, adjust=A, enhance=enhanced)))
        out["spa"], out["exp"] = np.array(float(loss.loss_spa(input=enhanced, target=x))), np.array(float(loss.loss_exp(input=enhanced)))
        out["col"], out["tva"] = np.array(float(loss.loss_col(input=enhanced))), np.array(float(loss.loss_tva(input=A)))
        out["weights"] = np.array([loss.spa_weight, loss.exp_weight, loss.col_weight, loss.tva_weight])
        # other constructor arguments: a coarser exposure grid and target, the 8-region spatial term, reduction "sum"
        out["exp_p8_e05"] = np.array(float(li["ExposureControlLoss"](patch_size=8, mean_val=0.5)(input=enhanced)))
        out["spa_r8"] = np.array(float(li["SpatialConsistencyLoss"](num_regions=8)(input=enhanced, target=x)))
        out["spa_sum"] = np.array(float(li["SpatialConsistencyLoss"](reduction="sum")(input=enhanced, target=x)))
        out["col_none"] = li["ColorConstancyLoss"](reduction="none")(input=enhanced).numpy()
        out["tva_w3"] = np.array(float(li["TotalVariationLoss"](loss_weight=3.0)(input=A)))
    path = os.path.join(HERE, "zdce_loss_mon_seed5_2x64x80.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB |", {k: float(out[k]) for k in ("total", "spa", "exp", "col", "tva")})


def gen_upsampler():
    """The PixelShuffle up-sampler leaf of the BASELINE config-3 decoder variant: Upsampler(default_conv, scale, n_feats, act=...) of
    src/mon/nn/modules/encoder/edsr.py:88-123 -- a file that imports torch only, loaded as a module."""
    mod = _load("src/mon/nn/modules/encoder/edsr.py", "ref_edsr")
    out = {}
    for tag, (nf, scale, act, hw, seed) in {"s2_c16": (16, 2, False, (9, 13), 111), "s4_c8_relu": (8, 4, "relu", (6, 7), 112), "s2_c32_relu": (32, 2, "relu", (5, 20), 113)}.items():
        torch.manual_seed(seed)
        up = mod.Upsampler(mod.default_conv, scale, nf, bn=False, act=act)
        x = torch.randn(2, nf, *hw, generator=torch.Generator().manual_seed(seed + 1000))
        with torch.no_grad():
            y = up(x)
        out.update({f"{tag}.w.{k}": v.numpy() for k, v in up.state_dict().items()})
        out[f"{tag}.x"], out[f"{tag}.y"], out[f"{tag}.scale"], out[f"{tag}.relu"] = x.numpy(), y.numpy(), np.array(scale), np.array(int(act == "relu"))
    path = os.path.join(HERE, "upsampler_edsr.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["zero_dce", "zero_dcepp", "llunetpp", "hinet", "zero_ig", "unet", "rrdnet", "retinexnet", "ssim", "zero_dce_losses", "upsampler", "hvi", "lin", "mprnet", "lllinet", "guided_filter", "mon_zero_dce_loss"]
    if "upsampler" in which:
        gen_upsampler()
    if "hvi" in which:
        gen_hvi()
    if "lin" in which:
        gen_learnable_instance_norm()
    if "mprnet" in which:
        gen_mprnet()
    if "lllinet" in which:
        gen_lllinet()
    if "guided_filter" in which:
        gen_guided_filter()
    if "mon_zero_dce_loss" in which:
        gen_mon_zero_dce_loss()
    if "retinexnet" in which:
        gen_retinexnet()
    if "ssim" in which:
        gen_ssim()
    if "zero_dce_losses" in which:
        gen_zero_dce_losses()
    if "zero_dce" in which:
        gen_zero_dce()
    if "zero_dcepp" in which:
        gen_zero_dcepp()
    if "llunetpp" in which:
        gen_llunetpp()
    if "hinet" in which:
        gen_hinet()
    if "zero_ig" in which:
        gen_zero_ig()
    if "unet" in which:
        gen_unet()
    if "rrdnet" in which:
        gen_rrdnet()
