"""CPU: the oracle (torch restatement + plain-C anchor) against the reference-generated goldens."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import zero_dce as o_zdce
from tests.helpers import load_golden, zero_dce_goldens


@pytest.mark.parametrize("path", zero_dce_goldens(), ids=lambda p: p.split("zero_dce_")[-1][:-4])
def test_torch_restatement_matches_reference(path):
    sd, g = load_golden(path)
    out = o_zdce.forward(sd, torch.from_numpy(g["image"]))
    # same ops, same library, same machine class: bit-identical to the reference file's outputs
    assert np.array_equal(out["enhanced"].numpy(), g["enhanced"])
    assert np.array_equal(out["adjust"].numpy(), g["adjust"])
    assert list(out.keys())[-1] == "enhanced"                      # predict.py:97 pops the last key
    assert [k for k in out if k.startswith("adjust_")] == [f"adjust_{i}" for i in range(8)]
    assert np.array_equal(o_zdce.to_uint8_save_image(out["enhanced"]).numpy(), g["enhanced_u8"])
    # enhance_image_1 of the original network == curve after 4 iterations
    e1 = o_zdce.curve(torch.from_numpy(g["image"]), out["adjust"], num_iters=4)
    assert np.array_equal(e1.numpy(), g["enhance_image_1"])


@pytest.mark.parametrize("path", zero_dce_goldens(), ids=lambda p: p.split("zero_dce_")[-1][:-4])
def test_c_anchor_matches_reference(path):
    sd, g = load_golden(path)
    out = c_oracle.zero_dce_forward(sd, g["image"])
    assert np.abs(out["enhanced"] - g["enhanced"]).max() < 5e-6
    assert np.abs(out["adjust"] - g["adjust"]).max() < 5e-6
    mism = (c_oracle.to_u8_round(out["enhanced"]) != g["enhanced_u8"])
    assert mism.mean() < 1e-3 and np.abs(c_oracle.to_u8_round(out["enhanced"]).astype(int) - g["enhanced_u8"].astype(int)).max() <= 1


def test_c_conv_variants_match_torch():
    """replicate padding / stride 2 / 9x9 / depthwise / 1x1-with-padding as used by RetinexNet and
    DSConv (retinexnet/model.py:26-66, dsconv.py:122-134)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 13, 17, generator=g)
    cases = [
        dict(co=6, k=9, s=1, p=4, mode="replicate", groups=1),
        dict(co=5, k=3, s=2, p=1, mode="replicate", groups=1),
        dict(co=4, k=3, s=1, p=1, mode="zeros", groups=4),
        dict(co=3, k=1, s=1, p=1, mode="replicate", groups=1),
        dict(co=7, k=4, s=2, p=1, mode="zeros", groups=1),
    ]
    for c in cases:
        w = torch.randn(c["co"], 4 // c["groups"], c["k"], c["k"], generator=g) * 0.2
        b = torch.randn(c["co"], generator=g)
        xp = F.pad(x, (c["p"],) * 4, mode="replicate") if c["mode"] == "replicate" else x
        ref = F.conv2d(xp, w, b, stride=c["s"], padding=0 if c["mode"] == "replicate" else c["p"], groups=c["groups"])
        got = c_oracle.conv2d(x.numpy(), w.numpy(), b.numpy(), stride=c["s"], pad=c["p"], pad_mode=c["mode"], groups=c["groups"])
        assert got.shape == tuple(ref.shape)
        assert np.abs(got - ref.numpy()).max() < 1e-5, c


def test_flop_count():
    assert o_zdce.FLOP_PER_PX == 158400
