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


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["zero_dce", "zero_dcepp", "llunetpp", "hinet", "zero_ig", "unet", "rrdnet"]
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
