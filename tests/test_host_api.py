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
    assert lib.ie_abi_version() == capi.ABI_VERSION


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


def test_padded_edge_row_maps_and_tape_bookkeeping():
    """Host logic of the 16-bit training graph that needs no GPU: the zero-padded weight scratch of channel-padded layers (per-source
    row maps, gradient scatter back into the master layout) and the tape's no-grad / real-channel bookkeeping."""
    from mon_amd import train as T
    taps, real, split, co, co_p = 2, [3, 3], [16, 8], 5, 8
    w = torch.arange(taps * sum(real) * co, dtype=torch.float32) + 1.0
    b = torch.arange(co, dtype=torch.float32) + 100.0
    e = T._PaddedEdge(w, b, taps, real, split, co, co_p)
    wp = e.w.view(taps, sum(split), co_p)
    ws = w.view(taps, sum(real), co)
    assert torch.equal(wp[:, 0:3, :co], ws[:, 0:3]) and torch.equal(wp[:, 16:19, :co], ws[:, 3:6])
    mask = torch.ones_like(wp, dtype=torch.bool)
    mask[:, 0:3, :co] = False
    mask[:, 16:19, :co] = False
    assert float(wp[mask].abs().max()) == 0.0                       # everything outside the two real row ranges / real couts is zero
    assert torch.equal(e.b[:co], b) and float(e.b[co:].abs().max()) == 0.0
    w += 1.0                                                        # master weights change (optimiser step) -> refresh picks them up
    e.refresh()
    assert torch.equal(e.w.view(taps, sum(split), co_p)[:, 16:19, :co], w.view(taps, sum(real), co)[:, 3:6])
    e.dw.copy_(torch.arange(e.dw.numel(), dtype=torch.float32))
    e.db.copy_(torch.arange(co_p, dtype=torch.float32))
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    e.scatter_grads(dw, db)
    gp = e.dw.view(taps, sum(split), co_p)
    assert torch.equal(dw.view(taps, sum(real), co)[:, 0:3], gp[:, 0:3, :co]) and torch.equal(dw.view(taps, sum(real), co)[:, 3:6], gp[:, 16:19, :co])
    assert torch.equal(db, e.db[:co])
    # tape: no-grad tensors have no slot, a second producer accumulates, real-channel marks default to the tensor's channels
    tape = T.Tape()
    x = torch.zeros(1, 16, 2, 2)
    tape.no_grad(x)
    assert tape.is_no_grad(x) and tape.slot(x) == (None, 0)
    y = torch.zeros(1, 16, 2, 2)
    buf, acc = tape.slot(y)
    assert acc == 0 and tape.slot(y)[1] == 1 and tape.grad(y) is buf
    assert T._real_c(y) == 16
    y._ie_real = 3
    assert T._real_c(y) == 3


def test_model_checkpoint_reference_naming(tmp_path):
    """nn/callback/model_checkpoint.py:35-185: <dir>/weights/<fullname>_best[_metric].ckpt (+ _last), each with a weights-only
    .pt twin; top-1 on the monitored value; get_latest_checkpoint picks the newest torch file (nn/model.py:79-92)."""
    import time
    import torch
    from mon_amd import train as T

    class FakeTrainer:
        step = 0

        class model:
            fullname = "zero_dce_re_sice_mix"

        def save_checkpoint(self, path, weights_only=False, epoch=0):
            d = {"state_dict": {"w": torch.zeros(1)}, "epoch": epoch, "global_step": self.step}
            if not weights_only:
                d["optimizer_states"] = [{}]
            os.makedirs(os.path.dirname(path), exist_ok=True)
            torch.save(d, path)

    tr = FakeTrainer()
    cb = T.ModelCheckpoint(tmp_path / "run", monitor="val/psnr", mode="max", save_last=True)
    assert cb.dirpath.endswith(os.path.join("run", "weights"))
    tr.step = 10
    w = cb.on_validation_end(tr, epoch=1, metrics={"val/psnr": 20.0})
    assert [os.path.basename(f) for f in w] == ["zero_dce_re_sice_mix_best_val_psnr.ckpt", "zero_dce_re_sice_mix_last.ckpt"]
    assert os.path.exists(w[0].replace(".ckpt", ".pt")) and "optimizer_states" not in torch.load(w[0].replace(".ckpt", ".pt"))
    tr.step = 20
    time.sleep(0.02)
    w = cb.on_validation_end(tr, epoch=2, metrics={"val/psnr": 19.0})            # worse: only `last` is rewritten
    assert [os.path.basename(f) for f in w] == ["zero_dce_re_sice_mix_last.ckpt"] and cb.best_model_score == 20.0
    latest = T.get_latest_checkpoint(tmp_path / "run")
    assert os.path.basename(latest).startswith("zero_dce_re_sice_mix_last") and T.get_global_step_from_checkpoint(latest) == 20
    assert T.get_epoch_from_checkpoint(cb.best_model_path) == 1 and cb.state_dict()["last_global_step_saved"] == 20
    assert T.get_latest_checkpoint(tmp_path / "nothing") is None
    assert T.ModelCheckpoint(tmp_path / "weights").dirpath == str(tmp_path / "weights")


def test_video_frame_sequence_round_trip(tmp_path):
    """video.py: the loader / writer protocol of core/data/dataset/video.py and core/video/io.py over PIL frame sequences (directory of
    frames, animated GIF); the cv2 classes are gated by the import, not faked."""
    import numpy as np
    from mon_amd import video, capi
    rng = np.random.default_rng(5)
    frames = [rng.integers(0, 256, (24, 40, 3), dtype=np.uint8) for _ in range(5)]
    w = video.FrameSequenceWriter(tmp_path / "clip", image_size=(24, 40), frame_rate=5)
    w.write(frames[0])
    w.write_batch(frames[1:3])
    w.write(torch.from_numpy(frames[3]).permute(2, 0, 1).float() / 255.0, denormalize=True)      # CHW float [0, 1]
    w.write(torch.from_numpy(frames[4]))                                                          # HWC uint8 tensor
    assert len(w) == 5
    w.close()
    ld = video.open_video(tmp_path / "clip")
    assert isinstance(ld, video.FrameSequenceLoader) and len(ld) == 5 and ld.shape == (24, 40, 3) and ld.imgsz == (24, 40)
    for i, d in enumerate(ld):
        assert np.array_equal(d["frame"], frames[i]) and d["meta"]["id"] == i
    got = list(ld.batches(2))
    assert [tuple(b.shape) for b, _ in got] == [(2, 24, 40, 3), (2, 24, 40, 3), (1, 24, 40, 3)] and got[0][0].dtype == torch.uint8
    t = video.FrameSequenceLoader(tmp_path / "clip", to_tensor=True)[2]["frame"]
    assert tuple(t.shape) == (3, 24, 40) and float(t.max()) <= 1.0
    # animated file + per-frame images
    flat = [np.full((8, 8, 3), 40 * i, dtype=np.uint8) for i in range(4)]
    g = video.FrameSequenceWriter(tmp_path / "anim.gif", image_size=(8, 8), frame_rate=10, save_image=True)
    g.write_batch(flat)
    g.close()
    ga = video.open_video(tmp_path / "anim.gif")
    assert len(ga) == 4 and sorted(p.name for p in (tmp_path / "anim").iterdir()) == [f"anim_{i:06d}.png" for i in range(4)]
    assert abs(int(ga[3]["frame"][0, 0, 0]) - 120) <= 2                   # GIF palette quantisation
    ga.close()
    with pytest.raises(IOError):
        video.FrameSequenceLoader(tmp_path / "missing.gif")
    with pytest.raises(ValueError):
        video._frame_u8("not a frame")
    # float frames: the explicit flag decides (to_image_nparray, utils.py:782-795), not the data -- a [0, 1] frame with a pixel at 1.01 stays bright
    f = np.full((2, 3, 3), 0.5, np.float32); f[0, 0, 0] = 1.01
    assert video._frame_u8(f, denormalize=True).max() == 255 and video._frame_u8(f, denormalize=True)[1, 1, 1] == 127
    assert video._frame_u8(f * 200.0).max() == 202 and video._frame_u8(np.array([[[300.0, -4.0, 7.9]]], np.float32)).tolist() == [[[255, 0, 7]]]
    # the ffmpeg writer's own rule (io.py:78-101): an ndarray is auto-detected with is_normalized_image whatever the flag says, a tensor follows the flag
    g01 = np.linspace(0.0, 1.0, 2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3)
    assert video._frame_u8_ffmpeg(g01).max() == 255 and video._frame_u8_ffmpeg(g01)[0, 1, 1] == int(g01[0, 1, 1] * 255.0)
    assert video._frame_u8_ffmpeg(g01 * 200.0).max() == 200                                   # not normalised: taken as [0, 255]
    chw = np.zeros((3, 4, 5), np.float32); chw[1] = 0.5
    assert video._frame_u8_ffmpeg(chw).shape == (4, 5, 3) and video._frame_u8_ffmpeg(chw)[2, 2].tolist() == [0, 127, 0]
    tt = torch.full((3, 4, 5), 0.5)
    assert video._frame_u8_ffmpeg(tt).max() == 0 and video._frame_u8_ffmpeg(tt, denormalize=True).max() == 127
    class Pipe:
        def __init__(self): self.buf = b""
        def write(self, b): self.buf += b
        def close(self): pass
    class Proc:
        stdin = Pipe()
        def wait(self): pass
    fw = video.VideoWriterFFmpeg.__new__(video.VideoWriterFFmpeg)       # no ffmpeg binary in this image: drive write() over a stand-in pipe
    fw.dst, fw.denormalize, fw.index, fw.save_image, fw.ffmpeg_process = tmp_path, False, 0, False, Proc()
    fw.image_size, fw.pix_fmt, fw.ffmpeg_kwargs, fw.verbose = (2, 3), "yuv420p", {"crf": 18}, False
    fw.write(g01)
    assert len(Proc.stdin.buf) == 2 * 3 * 3 and max(Proc.stdin.buf) == 255 and len(fw) == 1
    cmd = fw.command(tmp_path / "result.mp4")
    assert cmd[:8] == ["ffmpeg", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", "3x2", "-i"] and "-crf" in cmd and cmd[-1] == "-y"
    fw.close()
    import shutil
    if shutil.which("ffmpeg") is None:
        with pytest.raises(capi.IEError):
            video.VideoWriterFFmpeg(tmp_path / "out.mp4", image_size=(2, 3))
    # batches(): front to back through get_frame until the source ends -- also for a source that does not know its length (a stream)
    got = [b for b, _ in ld.batches(2)]
    assert [int(b.shape[0]) for b in got] == [2, 2, 1] and got[0].dtype == torch.uint8
    class Stream(video.VideoLoader):                                   # what VideoLoaderCV is for a "://" source: no length, frames until read() fails
        is_stream = True

        def get_data(self):
            self.num_frames, self.frame_height, self.frame_width = -1, 24, 40

        def get_frame(self, index):
            if index >= len(frames):
                raise IndexError(index)
            return frames[index]
    st = Stream(tmp_path / "clip")
    assert st.imgsz == (24, 40) and sum(int(b.shape[0]) for b, _ in st.batches(4)) == 5
    try:
        import cv2  # noqa: F401
    except ImportError:
        with pytest.raises(capi.IEError):
            video.VideoLoaderCV(tmp_path / "x.mp4")
        with pytest.raises(capi.IEError):
            video.VideoWriterCV(tmp_path / "y.mp4")
