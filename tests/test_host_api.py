"""CPU: registry / model host logic mirrors the reference's operator API (no GPU, no compute)."""
import os
import re

import pytest
import torch

import mon_amd
from mon_amd import capi
from mon_amd.registry import Factory, ModelFactory, depascalize, kebabize, pascalize
from tests.helpers import load_golden, zero_dce_goldens, ROOT


def test_name_conversions():
    assert kebabize("zero_dce_re") == "zero-dce-re"
    assert pascalize("zero_dce_re") == "ZeroDceRe"
    assert depascalize("ZeroDceRe") == "zero_dce_re"


def test_registry_register_and_build():
    reg = ModelFactory("Models")

    @reg.register(name="toy_net", arch="toy")
    class ToyNet:
        def __init__(self, a=1, **kw):
            self.a, self.kw, self.name = a, kw, None

    assert reg.archs == ["toy"] and reg.models == ["toy_net"]
    m = reg.build(name="toy_net", a=3)
    assert m.a == 3 and m.name == "toy_net"
    m = reg.build(config={"name": "toy-net", "a": 5, "extra": 1})   # kebab spelling + config dict
    assert m.a == 5 and m.kw == {"extra": 1}
    assert reg.build(name="ToyNet").a == 1                           # pascal spelling
    assert reg.build() is None and reg.build(config={"a": 1}) is None  # factory.py:351-355
    with pytest.raises(KeyError):
        reg.build(name="missing")
    with pytest.raises(TypeError):
        reg.register(name=3)
    with pytest.raises(ValueError):
        Factory("")
    objs = reg.build_instances([{"name": "toy_net", "a": 2}, "toy_net"])
    assert [o.a for o in objs] == [2, 1]


def test_models_registry_has_reference_names():
    assert "zero_dce_re" in mon_amd.MODELS.models and "zero_dce" in mon_amd.MODELS.archs


def test_zero_dce_host_contract():
    m = mon_amd.MODELS.build(config={"name": "zero_dce_re", "in_channels": 3, "num_channels": 32, "num_iters": 8,
                                     "root": "/tmp/x", "fullname": "zero_dce_re_test", "verbose": False,
                                     "loss": None, "metrics": None, "optimizers": None}, seed=7)
    assert m.name == "zero_dce_re" and m.params == 79416 and m.out_channels == 24
    sd = m.state_dict()
    assert list(sd)[:2] == ["e_conv1.weight", "e_conv1.bias"] and len(sd) == 14
    assert tuple(sd["e_conv7.weight"].shape) == (24, 64, 3, 3)
    assert abs(float(sd["e_conv2.weight"].std()) - 0.02) < 2e-3       # zero_dce.py:155
    with pytest.raises(ValueError):                                    # base.py:31-34
        m.forward({"not_image": torch.rand(1, 3, 8, 8)})
    with pytest.raises(capi.IEError):                                  # no CPU execution path
        m.forward({"image": torch.rand(1, 3, 8, 8)})


def test_state_dict_loading_rules(tmp_path):
    sd, _ = load_golden(zero_dce_goldens()[0])
    m = mon_amd.ZeroDCE_RE(weights=dict(sd))
    assert torch.equal(m.state_dict()["e_conv5.weight"], sd["e_conv5.weight"])
    # mon checkpoints carry loss.* tensors and may nest under "state_dict" 
    ck = dict(sd)
    ck["loss.loss_spa.weight_left"] = torch.zeros(1, 1, 3, 3)
    p = tmp_path / "zero_dce.pt"
    torch.save({"state_dict": ck}, p)
    m2 = mon_amd.MODELS.build(name="zero_dce_re", weights=str(p))
    assert torch.equal(m2.state_dict()["e_conv1.bias"], sd["e_conv1.bias"])
    # strict on conv keys / shapes
    bad = dict(sd); bad.pop("e_conv3.bias")
    with pytest.raises(RuntimeError, match="Missing key"):
        mon_amd.ZeroDCE_RE().load_state_dict(bad)
    bad = dict(sd); bad["e_conv9.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        mon_amd.ZeroDCE_RE().load_state_dict(bad)
    bad = dict(sd); bad["e_conv7.weight"] = torch.zeros(3, 64, 3, 3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        mon_amd.ZeroDCE_RE().load_state_dict(bad)
    # a .ckpt path is rejected like the reference (nn/model.py:401-405)
    q = tmp_path / "zero_dce.ckpt"
    torch.save(ck, q)
    with pytest.raises(ValueError):
        mon_amd.ZeroDCE_RE(weights=str(q))


def test_capi_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports every function include/ie_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "ie_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ie_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    lib = capi.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} not exported by {capi.LIB_PATH}"
    assert lib.ie_abi_version() == 1


def test_sharding_ranges():
    from mon_amd.sharding import shard_range
    for n in (0, 1, 7, 64, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)
