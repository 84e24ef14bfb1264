"""Training path of the hot path: forward with saved activations -> loss -> hand-written backward
kernels -> gradient all-reduce (RCCL) -> global-norm clip -> Adam, without torch.autograd.

Reference: ``ImageEnhancementModel.forward_loss`` (src/mon/vision/enhance/base.py:47-57), ``Model.training_step``
(src/mon/nn/model.py:701-746), ``configure_optimizers`` (:498-615; optimiser parameter filter nn/factory.py:66-80),
Adam + ``gradient_clip_val`` (llie/zero_dce/config/zero_dce_re_sice_mix.py:46-51,106), Lightning DDP = mean of the
per-rank gradients.

Design: all trainable parameters live in ONE flat fp32 device buffer (conv weights in the kernels' HWIO layout), with
flat gradient / Adam-moment buffers beside it: one all-reduce, one norm reduction and one Adam launch per step.
``Tape`` is a minimal reverse-mode recorder over the ops of ``ops.py``; every backward op is one C-ABI call.
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from . import capi, ops
from .losses import LOSSES, Loss

CL = torch.channels_last


def _os_env(name, default):
    import os
    return os.environ.get(name, default)


class Tape:
    """Reverse-mode recorder.  Gradient buffers are created on first use; a second producer accumulates."""

    def __init__(self, recording=True):
        # recording=False: a forward-only tape (a sub-network that runs inside a training forward but is not trained in this phase).  Its backward closures
        # are dropped on the spot: kept, they would reference the tape that holds them -- a cycle nothing ever breaks, because backward() is never called on
        # such a tape -- and every activation of that sub-network would live until Python's cycle collector happens to run (RetinexNet, round 3: +12 GB of
        # reserved memory every second step, 0.4 s segment allocations, 152 GB after 18 steps).
        self._fns, self._g, self._nograd, self._recording = [], {}, set(), bool(recording)

    def no_grad(self, t):
        self._nograd.add(id(t))
        self._g[id(t)] = (t, None)
        return t

    def is_no_grad(self, t):
        return id(t) in self._nograd

    def slot(self, t):
        """(gradient buffer of ``t`` or None, accumulate flag) for a producer about to write into it."""
        k = id(t)
        if k in self._nograd:
            return None, 0
        if k in self._g:
            return self._g[k][1], 1
        buf = torch.empty_like(t, memory_format=CL)
        self._g[k] = (t, buf)
        return buf, 0

    def seed(self, t, grad):
        self._g[id(t)] = (t, grad)

    def grad(self, t):
        e = self._g.get(id(t))
        return None if e is None else e[1]

    def release(self, t):
        """Drop the gradient buffer of ``t`` once its producer has consumed it."""
        self._g.pop(id(t), None)

    def record(self, fn):
        if self._recording:
            self._fns.append(fn)

    def mark_grad_bucket(self, offset: int):
        """Forward-order mark: every parameter used AFTER this point lives at flat-buffer offsets >= ``offset``.  When the backward pass
        comes back to the mark those gradients are final, and ``on_grad_bucket`` (the trainer's early all-reduce) is called with it."""
        def fire():
            if self.on_grad_bucket is not None:
                self.on_grad_bucket(int(offset))
        self._fns.append(fire)

    on_grad_bucket = None
    # Weight gradients off the critical path (Trainer.overlap_wgrad): a layer's weight gradient has no consumer before the optimiser step, while its data gradient
    # heads the chain the rest of the backward pass waits for.  With a side stream set, ``on_side`` enqueues the weight-gradient launches there (behind everything
    # the current stream has enqueued so far); the matrix-core weight-gradient kernels then run beside the HBM-bound BatchNorm / pooling / resize passes of the
    # layers in front.  ``join_side`` makes the current stream wait for them (before the gradient exchange and at the end of the pass).
    wgrad_stream = None
    _side_used = False

    def on_side(self, tensors, fn):
        side = self.wgrad_stream
        if side is None:
            fn()
            return
        side.wait_stream(torch.cuda.current_stream(side.device))
        with torch.cuda.stream(side):
            fn()
        for t in tensors:                           # the caching allocator must not hand their memory out again before the side stream is done with it
            if t is not None:
                t.record_stream(side)
        self._side_used = True

    def join_side(self):
        if self._side_used:
            torch.cuda.current_stream(self.wgrad_stream.device).wait_stream(self.wgrad_stream)
            self._side_used = False

    def backward(self):
        for fn in reversed(self._fns):
            fn()
        self.join_side()
        self._fns = []
        self._g, self._nograd = {}, set()          # gradient buffers and their tensors: nothing of a finished step stays reachable through the tape


class ParamStore:
    """Flat device storage of a model's trainable parameters (+ gradient and Adam moments) and its buffers."""

    def __init__(self, model, device, compute_dtype="f32"):
        self.device = device
        self.compute_dtype = compute_dtype        # "f32", or "bf16"/"f16": activations between the convolutions are 16-bit
        self._c16 = {}                            # (name, split) -> (forward MConvWeightsDev, [data-gradient MConvWeightsDev per source])
        self.index = OrderedDict()            # name -> (offset, numel, shape, is_conv)
        self.buffers = OrderedDict()          # running_mean / running_var
        # ConvTranspose2d(k=2, s=2) weights (Cin, Cout, 2, 2) are kept as they are: read as [Cin][4*Cout] that IS the HWIO matrix of
        # the 1x1 conv to channels o*4 + i*2 + j that, followed by a pixel shuffle, equals the transposed conv
        self.convT = set(getattr(model, "conv_transpose_keys", lambda: [])())
        off = 0
        for k, v in model._params.items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                self.buffers[k] = v.detach().to(device, torch.float32).contiguous().clone()
                continue
            if any(s in k for s in ("loss", "train/", "val/", "test/")):      # nn/factory.py:70-77
                continue
            self.index[k] = (off, v.numel(), tuple(v.shape), v.dim() == 4 and k not in self.convT)
            off += (v.numel() + 3) // 4 * 4                                    # keep every view 16-byte aligned
        self.numel = off
        self.params = torch.zeros(off, device=device, dtype=torch.float32)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        for k, (o, n, shape, is_conv) in self.index.items():
            if is_conv:
                ops.pack_hwio_into(model._params[k], self.params[o:o + n])
            else:
                self.params[o:o + n].copy_(model._params[k].reshape(-1))

    def view(self, name, of=None):
        o, n, _, _ = self.index[name]
        return (self.params if of is None else of)[o:o + n]

    def conv(self, name) -> ops.ConvWeightsView:
        _, _, (co, ci, kh, kw), _ = self.index[f"{name}.weight"]
        b = self.view(f"{name}.bias") if f"{name}.bias" in self.index else None
        cw = ops.ConvWeightsView(self.view(f"{name}.weight"), b, co, ci, kh, kw)
        cw.store, cw.name = self, name
        return cw

    def conv16(self, name, split, cout_pad=None, real_split=None):
        """Packed 16-bit weights of conv ``name`` over sources ``split`` for the matrix-core forward and for the data gradient of
        every source, produced on the device from the fp32 master weights and refreshed by ``repack16()`` after every step.

        Channel-padded edges (RGB carried as 16 channels, a 3-channel head computed as 16): ``real_split`` gives the channels of
        each source that exist in the layer's weight matrix (the rest of the source is zero padding) and ``cout_pad`` the padded
        Cout; the packers then read a zero-padded fp32 scratch copy of the master weights (``pad`` in the returned triple: scratch
        weights / bias and their gradient twins), refreshed before every repack."""
        split = [int(c) for c in split]
        real = [int(c) for c in (real_split or split)]
        key = (name, tuple(split), cout_pad, tuple(real))
        if key not in self._c16:
            _, _, (co, ci, kh, kw), _ = self.index[f"{name}.weight"]
            dt = ops.TORCH_DT[self.compute_dtype]
            b = self.view(f"{name}.bias") if f"{name}.bias" in self.index else None
            if f"{name}.weight" in self.convT:       # (Cin, Cout, 2, 2) read as the HWIO matrix of a 1x1 conv Cin -> 4*Cout; its bias is the expanded one
                ci, co, kh, kw = co, 4 * ci, 1, 1          # (the index holds (Cin, Cout, 2, 2), unpacked above as (co, ci, kh, kw))
                b = self.conv_transpose_bias4(name) if b is not None else None
            w = self.view(f"{name}.weight")
            co_p, pad = cout_pad or co, None
            if sum(real) != ci or co_p < co or any(r > c for r, c in zip(real, split)):
                raise capi.IEError(f"conv16 {name}: sources {split} (real {real}) / cout {co_p} do not fit the layer {ci}->{co}")
            if real != split or co_p != co:
                pad = _PaddedEdge(w, b, kh * kw, real, split, co, co_p)
                w, b = pad.w, pad.b
            fwd = ops.MConvWeightsDev(w, b, split, co_p, kh, kw, dt)
            dg = [ops.MConvWeightsDev(w, None, split, co_p, kh, kw, dt, mode="dgrad", src_index=i) for i in range(len(split))]
            self._c16[key] = (fwd, dg, pad)
        return self._c16[key]

    def repack16(self):
        for fwd, dg, pad in self._c16.values():
            if pad is not None:
                pad.refresh()
            fwd.repack()
            for g in dg:
                g.repack()

    def conv_transpose(self, name):
        """(weights view of the equivalent 1x1 conv, its weight-gradient view) of ConvTranspose2d ``name``; the bias (Cout) is tied
        over the 4 sub-pixels: ``expand_bias`` / ``reduce_bias_grad`` move between the (Cout) parameter and the 4*Cout conv bias."""
        _, n, (ci, co, kh, kw), _ = self.index[f"{name}.weight"]
        w = ops.ConvWeightsView(self.view(f"{name}.weight"), None, co * 4, ci, 1, 1)
        w.store, w.name = self, name          # the 16-bit graph packs it through conv16 (the 1x1 conv to 4*Cout channels)
        return w, self.view(f"{name}.weight", self.grads)

    def conv_transpose_bias4(self, name):
        """The (4*Cout) bias of the 1x1 conv that stands for ConvTranspose2d ``name``: a persistent buffer (the packed 16-bit forward keeps a pointer
        to it), refreshed from the (Cout) parameter at every forward."""
        if not hasattr(self, "_ct_bias"):
            self._ct_bias = {}
        b = self.view(f"{name}.bias").repeat_interleave(4)
        if name not in self._ct_bias:
            self._ct_bias[name] = b.contiguous().clone()
        else:
            self._ct_bias[name].copy_(b)
        return self._ct_bias[name]

    def conv_grads(self, name):
        db = self.view(f"{name}.bias", self.grads) if f"{name}.bias" in self.index else None
        return self.view(f"{name}.weight", self.grads), db

    def export(self, of=None) -> "OrderedDict[str, torch.Tensor]":
        """Checkpoint-layout (OIHW) host copies of the parameters (or of ``of`` = grads / moments), plus buffers."""
        out = OrderedDict()
        for k, (o, n, shape, is_conv) in self.index.items():
            t = (self.params if of is None else of)[o:o + n]
            if is_conv:
                co, ci, kh, kw = shape
                t = t.reshape(kh, kw, ci, co).permute(3, 2, 0, 1)
            out[k] = t.reshape(shape).to("cpu").contiguous().clone()
        if of is None:
            for k, v in self.buffers.items():
                out[k] = v.to("cpu").clone()
        return out


class _PaddedEdge:
    """Zero-padded fp32 scratch copy [taps][sum(split)][co_p] of a conv's master weights [taps][sum(real)][co] (and bias), plus
    gradient scratch of the same shape: source s owns rows [sum(split[:s]), +real[s]) of the padded matrix.  Device-to-device
    slice copies only, the arithmetic stays in the kernels."""

    def __init__(self, w, b, taps, real, split, co, co_p):
        self.src_w, self.src_b, self.taps, self.co, self.co_p = w, b, taps, co, co_p
        self.ci, self.cin_p = sum(real), sum(split)
        self.rows, r0, p0 = [], 0, 0
        for r, c in zip(real, split):
            self.rows.append((p0, r0, r))
            r0 += r; p0 += c
        self.w = torch.zeros(taps * self.cin_p * co_p, device=w.device, dtype=torch.float32)
        self.b = None if b is None else torch.zeros(co_p, device=w.device, dtype=torch.float32)
        self.dw = torch.zeros_like(self.w)
        self.db = None if b is None else torch.zeros_like(self.b)
        self.refresh()

    def refresh(self):
        wp, ws = self.w.view(self.taps, self.cin_p, self.co_p), self.src_w.view(self.taps, self.ci, self.co)
        for p0, r0, r in self.rows:
            wp[:, p0:p0 + r, :self.co].copy_(ws[:, r0:r0 + r, :])
        if self.b is not None:
            self.b[:self.co].copy_(self.src_b)

    def scatter_grads(self, dw, db):
        gp, gs = self.dw.view(self.taps, self.cin_p, self.co_p), dw.view(self.taps, self.ci, self.co)
        for p0, r0, r in self.rows:
            gs[:, r0:r0 + r, :].copy_(gp[:, p0:p0 + r, :self.co])
        if db is not None:
            db.copy_(self.db[:self.co])


def _real_c(t):
    """Channels of ``t`` that carry data (16-bit tensors of the padded edges hold zeros behind them)."""
    return int(getattr(t, "_ie_real", t.shape[1]))


# ---- taped ops ----# ---- taped ops ---------------------------------------------------------------------------------------------------------
def t_cast(tape, x, dtype, channels=None):
    """dtype (and channel count) change between the fp32 edges and the 16-bit interior of the mixed-precision graph."""
    y = ops.cast(x, dtype, channels)
    if y.shape[1] > x.shape[1] or _real_c(x) < x.shape[1]:
        y._ie_real = min(_real_c(x), int(y.shape[1]))
    if tape.is_no_grad(x):
        return tape.no_grad(y)

    def bwd():
        g = tape.grad(y)
        if g is None or tape.is_no_grad(x):
            return
        gx = ops.cast(g, x.dtype, x.shape[1])
        if tape.grad(x) is not None:
            _accumulate(tape.grad(x), gx)
        else:
            tape.seed(x, gx)
        tape.release(y)
    tape.record(bwd)
    return y


def t_conv16(tape, srcs, cw, dw, db, pad=1, cout_pad=None, act="none", slope=0.0, stride=1):
    """16-bit conv (stride 1 or 2, zero padding, activation fused into the forward epilogue): forward, data gradient and (3x3 stride 1) weight
    gradient on the matrix cores, weights packed on the device from the fp32 master copy; fp32 gradient accumulation.
    ``cout_pad`` / sources wider than the layer's Cin: channel-padded edges (``ParamStore.conv16``).  Stride 2: the data gradient is the
    stride-1 convolution of the zero-inserted output gradient (``ops.zero_insert2``) with the same flipped weights."""
    fwd, dgw, edge = cw.store.conv16(cw.name, [int(s.shape[1]) for s in srcs], cout_pad, [_real_c(s) for s in srcs])
    y = ops.mconv2d(srcs, fwd, stride=stride, pad=pad, act=act, slope=slope)
    if cout_pad and cout_pad != cw.cout:
        y._ie_real = cw.cout

    def bwd():
        dy = tape.grad(y)
        if dy is None:
            return
        if act != "none":
            ops.act_backward_into(dy, y, dy, act, slope)          # in place: this op owns the gradient buffer of its output
        def wgrad():
            if edge is None:
                ops.conv2d_wgrad(srcs, dy, cw, dw, db, stride, pad, "zeros")
            else:
                ev = ops.ConvWeightsView(edge.w, edge.b, fwd.cout, sum(fwd.split), cw.kh, cw.kw)
                ops.conv2d_wgrad(srcs, dy, ev, edge.dw, edge.db if db is not None else None, stride, pad, "zeros")
                edge.scatter_grads(dw, db)
        tape.on_side(list(srcs) + [dy], wgrad)      # (dy is final here: this op owns the gradient buffer of its output and nothing writes it again)
        dyz = None
        for i, src in enumerate(srcs):
            if tape.is_no_grad(src):
                continue
            if stride == 2 and dyz is None:          # (H + 2 pad - k + 1) rows: the stride-1 full convolution of it has the source's H rows
                dyz = ops.zero_insert2(dy, int(src.shape[2]) + 2 * pad - cw.kh + 1, int(src.shape[3]) + 2 * pad - cw.kw + 1)
            g = ops.mconv2d([dy if stride == 1 else dyz], dgw[i], pad=cw.kh - 1 - pad)
            if tape.grad(src) is not None:
                _accumulate(tape.grad(src), g)
            else:
                tape.seed(src, g)
        tape.release(y)
    tape.record(bwd)
    return y


def t_conv(tape, srcs, cw, dw, db, stride=1, pad=1, pad_mode="zeros", act="none", slope=0.0, cout_pad=None):
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    if srcs[0].dtype != torch.float32:
        if stride not in (1, 2) or pad_mode != "zeros":
            raise capi.IEError("the 16-bit training conv supports stride 1 / 2 and zero padding (replicate padding: t_pad_replicate in front of it)")
        return t_conv16(tape, srcs, cw, dw, db, pad, cout_pad, act, slope, stride)
    y = ops.conv2d(srcs, cw, stride=stride, pad=pad, pad_mode=pad_mode, act=act, slope=slope)

    def bwd():
        dy = tape.grad(y)
        if dy is None:
            return
        if act != "none":
            g = torch.empty_like(dy, memory_format=CL)
            ops.act_backward_into(dy, y, g, act, slope)
            dy = g
        ops.conv2d_wgrad(srcs, dy, cw, dw, db, stride, pad, pad_mode)
        slots = [tape.slot(s) for s in srcs]
        if any(b is not None for b, _ in slots):
            ops.conv2d_dgrad(dy, cw, [tuple(s.shape) for s in srcs], [b for b, _ in slots], [a for _, a in slots], stride, pad, pad_mode)
        tape.release(y)
    tape.record(bwd)
    return y


def t_norm(tape, x, gamma, beta, dgamma, dbeta, per_sample, running_mean=None, running_var=None, momentum=0.1, eps=1e-5,
           act="none", slope=0.0, c_begin=0, c_count=None):
    if c_count is None and _real_c(x) < x.shape[1]:        # channel-padded 16-bit tensor: normalise the real channels only
        c_count = _real_c(x) - c_begin
    y, mean, inv = ops.norm_forward(x, gamma, beta, per_sample, running_mean, running_var, momentum, eps, act, slope, c_begin, c_count)
    if _real_c(x) < x.shape[1]:
        y._ie_real = _real_c(x)

    def bwd():
        dy = tape.grad(y)
        if dy is None:
            return
        # none / ReLU / LeakyReLU: the activation's derivative from the recomputed pre-activation -- the two passes read x and dy, not y (IE_NORM_RECOMPUTE=0: A/B)
        pre = act in ("none", "relu", "lrelu") and _os_env("IE_NORM_RECOMPUTE", "1") != "0"
        dx = ops.norm_backward(x, dy, y, gamma, mean, inv, per_sample, dgamma, dbeta, False, act, slope, c_begin, c_count, beta=beta, recompute=pre)
        if tape.is_no_grad(x):
            return
        if tape.grad(x) is not None:
            _accumulate(tape.grad(x), dx)
        else:
            tape.seed(x, dx)
        tape.release(y)
    tape.record(bwd)
    return y


def _accumulate(buf, dx):
    """buf += dx through the library's eltwise kernel (no torch arithmetic on the data path)."""
    n, c, h, w = buf.shape
    if buf.dtype != torch.float32:
        with torch.cuda.device(buf.device):
            capi.check(capi.lib().ie_eltwise_nhwc(0, capi.ptr(buf), capi.ptr(dx), capi.ptr(buf), ops._dt(buf), n * h * w, c, c,
                                                  capi.current_stream()), "ie_eltwise_nhwc")
        return
    with torch.cuda.device(buf.device):
        capi.check(capi.lib().ie_eltwise_nhwc_f32(0, capi.ptr(buf), capi.ptr(dx), capi.ptr(buf), n * h * w, c, c, 0.0, 0.0,
                                                  capi.current_stream()), "ie_eltwise_nhwc_f32")


def t_maxpool(tape, x):
    y = ops.maxpool2x2(x)

    def bwd():
        dy = tape.grad(y)
        if dy is None:
            return
        buf, acc = tape.slot(x)
        if buf is not None:
            ops.maxpool2x2_backward(x, dy, buf, acc)
        tape.release(y)
    tape.record(bwd)
    return y


def t_resize(tape, x, size, mode="bilinear_align"):
    y = ops.resize(x, size, mode)

    def bwd():
        dy = tape.grad(y)
        if dy is None:
            return
        buf, acc = tape.slot(x)
        if buf is not None:
            ops.resize_backward(dy, buf, mode, acc)
        tape.release(y)
    tape.record(bwd)
    return y


def t_pad(tape, x, top, bottom, left, right):
    if top == bottom == left == right == 0:
        return x
    y = ops.pad(x, top, bottom, left, right)

    def bwd():
        dy = tape.grad(y)
        if dy is None:
            return
        buf, acc = tape.slot(x)
        if buf is not None:
            ops.crop_into(dy, buf, top, left, acc)
        tape.release(y)
    tape.record(bwd)
    return y


def t_pad_replicate(tape, x, pad):
    """Replicate padding as its own taped op: a ``padding_mode="replicate"`` convolution (retinexnet/model.py:26-66) is this followed by
    the pad-0 convolution, whose weight and data gradients are the zero-padding kernels' on the padded tensor."""
    if pad == 0:
        return x
    y = ops.pad_replicate(x, pad)
    if _real_c(x) < x.shape[1]:
        y._ie_real = _real_c(x)
    if tape.is_no_grad(x):
        return tape.no_grad(y)

    def bwd():
        dy = tape.grad(y)
        if dy is None:
            return
        buf, acc = tape.slot(x)
        if buf is not None:
            ops.pad_replicate_backward_into(dy, buf, pad, acc)
        tape.release(y)
    tape.record(bwd)
    return y


def t_pixel_shuffle(tape, x, r=2):
    y = ops.pixel_shuffle(x, r)

    def bwd():
        dy = tape.grad(y)
        if dy is None:
            return
        buf, acc = tape.slot(x)
        if buf is not None:
            ops.pixel_unshuffle_into(dy, buf, r, acc)
        tape.release(y)
    tape.record(bwd)
    return y


def t_clamp(tape, x, lo, hi):
    y = ops.clamp(x, lo, hi)

    def bwd():
        dy = tape.grad(y)
        if dy is None:
            return
        buf, acc = tape.slot(x)
        if buf is not None:
            ops.act_backward_into(dy, x, buf, "clamp", 0.0, lo, hi, acc)
        tape.release(y)
    tape.record(bwd)
    return y


def t_add(tape, a, b):
    """a + b (residual connection): the gradient goes to both operands unchanged."""
    y = ops.add(a, b)
    if max(_real_c(a), _real_c(b)) < y.shape[1]:       # channel-padded operands (RGB carried as 16): the sum is padded the same way
        y._ie_real = max(_real_c(a), _real_c(b))

    def bwd():
        g = tape.grad(y)
        if g is None:
            return
        for t in (a, b):
            if tape.is_no_grad(t):
                continue
            if tape.grad(t) is not None:
                _accumulate(tape.grad(t), g)
            elif t is a and not tape.is_no_grad(b) and tape.grad(b) is None:
                tape.seed(t, g.clone(memory_format=torch.preserve_format))   # the other operand will own `g` itself
            else:
                tape.seed(t, g)
        tape.release(y)
    tape.record(bwd)
    return y


def t_mul(tape, a, b):
    """a * b (same shape): ga = g * b, gb = g * a."""
    y = ops.mul(a, b)

    def bwd():
        g = tape.grad(y)
        if g is None:
            return
        for t, other in ((a, b), (b, a)):
            if tape.is_no_grad(t):
                continue
            gt = ops.mul(g, other)
            if tape.grad(t) is not None:
                _accumulate(tape.grad(t), gt)
            else:
                tape.seed(t, gt)
        tape.release(y)
    tape.record(bwd)
    return y


def t_conv_transpose2x2(tape, x, store, name):
    """nn.ConvTranspose2d(k=2, s=2) = 1x1 conv to 4*Cout channels (o*4 + i*2 + j) + pixel shuffle; the bias is tied over the 4
    sub-pixels (expanded for the forward, its gradient summed back)."""
    cw, dw = store.conv_transpose(name)
    has_b = f"{name}.bias" in store.index
    b4 = store.conv_transpose_bias4(name) if has_b else None     # parameter plumbing, Cout values
    cw.b = b4
    db4 = torch.empty_like(b4) if has_b else None
    y = t_pixel_shuffle(tape, t_conv(tape, x, cw, dw, db4, pad=0), 2)
    if has_b:
        def bwd():
            # (behind the layer's weight gradient, on the stream that ran it)
            tape.on_side([db4], lambda: store.view(f"{name}.bias", store.grads).copy_(db4.view(-1, 4).sum(1)))
        # the tape runs in reverse order of recording: placed BEFORE the conv node, the reduction executes right after it
        tape._fns.insert(len(tape._fns) - 2, bwd)
    return y


def t_curve(tape, image_nchw, adjust, iters):
    """LE curve (zero_dce.py:186-188): image NCHW fp32 (no gradient) + curve maps NHWC -> enhanced NCHW fp32."""
    n, _, h, w = image_nchw.shape
    cp = int(adjust.shape[1])
    y = torch.empty_like(image_nchw)
    L = capi.lib()
    with torch.cuda.device(image_nchw.device):
        capi.check(L.ie_le_curve(capi.ptr(image_nchw), capi.ptr(adjust), cp, capi.ptr(y), n, h, w, iters, capi.current_stream()), "ie_le_curve")

    def bwd():
        g = tape.grad(y)
        if g is None:
            return
        buf, acc = tape.slot(adjust)
        if buf is not None:
            with torch.cuda.device(g.device):
                capi.check(L.ie_le_curve_backward(capi.ptr(image_nchw), capi.ptr(adjust), cp, capi.ptr(g), capi.ptr(buf), n, h, w, iters, int(acc),
                                                  capi.current_stream()), "ie_le_curve_backward")
        tape.release(y)
    tape.record(bwd)
    return y


# ---- trainer -------------------------------------------------------------------------------------------------------------
class Trainer:
    """One-model fit driver: ``training_step(batch)`` = forward_loss + backward + all-reduce + clip + Adam.

    ``model`` must implement ``forward_train(image_nhwc, store: ParamStore, tape: Tape) -> enhanced (NHWC tensor)``.
    ``optimizer``: ``{"name": "adam", "lr": ..., "betas": (..), "eps": .., "weight_decay": ..}`` like the reference configs'
    ``optimizers[0]["optimizer"]``; ``gradient_clip_val`` like the Lightning trainer config (norm clipping).
    ``process_group``: torch.distributed group for data parallelism (default group when initialised).
    ``deterministic``: weight gradients through the two-stage fixed-order reduction (run-to-run bit-identical steps)."""

    def __init__(self, model, optimizer: dict | None = None, loss=None, gradient_clip_val: float | None = 0.1, process_group=None,
                 compute_dtype: str = "f32", loss_scale: float | None = None, deterministic: bool = False):
        if not hasattr(model, "forward_train") and not hasattr(model, "forward_train_loss"):
            raise capi.IEError(f"{model.__class__.__name__} has no training graph (forward_train)")
        model._require_cuda()
        self.model, self.device = model, model.device
        opt = dict(optimizer or {})
        name = opt.pop("name", "adam")
        if name.lower() != "adam":
            raise capi.IEError(f"optimizer {name!r} is not implemented (adam only)")
        self.lr = float(opt.pop("lr", 1e-4))
        self.betas = tuple(opt.pop("betas", (0.9, 0.999)))
        self.eps = float(opt.pop("eps", 1e-8))
        self.weight_decay = float(opt.pop("weight_decay", 0.0))
        self.clip = float(gradient_clip_val) if gradient_clip_val else 0.0
        if isinstance(loss, dict):
            loss = LOSSES.build(config=loss)
        from .globals_ import Scheme
        zero_ref = Scheme.ZERO_REFERENCE in getattr(model, "schemes", [])
        self.loss = loss if loss is not None else LOSSES.build(name="zero_dce_loss" if zero_ref else "l1_ssim_loss")
        if not isinstance(self.loss, Loss):
            raise TypeError("`loss` must be a mon_amd.losses.Loss or its config dict")
        self.group = process_group
        # data parallel step: the gradients behind the model's bucket mark (the decoder's -- their backward runs first) are all-reduced
        # on the collective's own stream while the rest of the backward pass (the encoder's data / weight gradients) still runs
        self.overlap_allreduce = True
        self.sync_gradients = True               # False: skip the exchange (timing the step without it)
        # 16-bit graphs: weight-gradient launches on a side stream, beside the HBM-bound passes of the layers in front (Tape.on_side).  Off in deterministic mode (the
        # fixed-order reduction shares one library workspace per device) and with IE_WGRAD_STREAM=0 (A/B).
        import os as _os
        self.overlap_wgrad = (not deterministic) and _os.environ.get("IE_WGRAD_STREAM", "1") != "0"
        self._wgrad_stream = None
        self._pending = []                        # [(async work handle)], first bucket offset
        self._bucket_off = None
        # process-wide switch of the library (capi.set_deterministic): turned on here, restored by ``close()`` (or use the trainer as a context manager)
        self._prev_deterministic = None
        if deterministic:
            self._prev_deterministic = capi.set_deterministic(True)      # returns the previous setting
        cd = {capi.IE_F32: "f32", capi.IE_F16: "f16", capi.IE_BF16: "bf16"}[capi.DTYPES[str(compute_dtype).lower()]]
        if cd != "f32" and not getattr(model, "supports_mixed_training", False):
            raise capi.IEError(f"{model.__class__.__name__} has no 16-bit training graph")
        self.compute_dtype = cd
        # Static loss scale (AMP): the seed gradient is multiplied by it and the optimiser divides it out again.  fp16 needs it --
        # a mean-reduced loss over a 1080p batch seeds ~1e-7 per element, below fp16's normal range; bf16 / fp32 do not.  With
        # clipping on, a step whose scaled gradients overflowed (non-finite norm) is skipped by ie_adam_step_f32.
        self.loss_scale = float(loss_scale) if loss_scale else (8192.0 if cd == "f16" else 1.0)
        if getattr(self.loss, "zero_reference", False):
            self.loss.grad_scale = self.loss_scale          # this loss computes its gradients in the forward pass
        self.store = ParamStore(model, self.device, cd)
        # {step, applied, bc1, bc2_sqrt} on the device: a skipped (overflowed) step does not advance Adam's step count
        self._opt_state = torch.zeros(4, device=self.device, dtype=torch.int32)
        self._sumsq = torch.zeros(1, device=self.device, dtype=torch.float64)
        self._sumsq_ws = torch.zeros(capi.SUMSQ_WS_DOUBLES, device=self.device, dtype=torch.float64)
        model.predicting = False
        model.train()

    def close(self):
        """Give the library's process-wide deterministic switch back to what it was before this trainer turned it on."""
        if self._prev_deterministic is not None:
            capi.set_deterministic(bool(self._prev_deterministic))
            self._prev_deterministic = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @property
    def step(self) -> int:
        """Optimiser steps actually applied (a host read of the device-side counter; checkpointed as ``global_step``)."""
        return int(self._opt_state[0].item())

    @step.setter
    def step(self, v: int):
        self._opt_state[0] = int(v)

    @property
    def last_step_applied(self) -> bool:
        return bool(self._opt_state[1].item())

    # nn/model.py:498 -- kept for API parity; the optimiser state lives in the ParamStore
    def configure_optimizers(self):
        return {"optimizer": {"name": "adam", "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}}

    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def forward_loss(self, datapoint: dict, tape: Tape | None = None) -> dict:
        """vision/enhance/base.py:47-57."""
        m = self.model
        m.assert_datapoint(datapoint)
        tape = tape or Tape()
        if hasattr(m, "forward_train_loss"):      # models whose training forward computes its own losses (retinexnet/model.py:199-238)
            outputs = m.forward_train_loss(datapoint, self.store, tape)
            outputs["_tape"] = tape
            return outputs
        x = ops.to_nhwc(datapoint["image"].to(self.device, torch.float32))
        tape.no_grad(x)
        res = m.forward_train(x, self.store, tape)
        if isinstance(res, dict) and "adjust" not in res:
            # supervised models with several predictions (hinet.py:311-320): loss = sum over the outputs of loss(p, target)
            outputs = OrderedDict(res)
            m.assert_outputs(outputs)
            target = datapoint.get("ref_image").to(self.device, torch.float32)
            parts = [(p_, self.loss(p_, target)) for p_ in outputs.values()]

            def seed(tape_, scale=1.0):
                for p_, r_ in parts:
                    g_ = torch.empty_like(p_, memory_format=CL)
                    r_.backward_into(g_, False, scale)
                    tape_.seed(p_, g_)
            from .losses import LossResult
            outputs["loss"] = LossResult(lambda: sum(float(r_) for _, r_ in parts), None, seed)
        elif isinstance(res, dict):               # zero-reference models: loss(input, adjust, enhance) (zero_dce.py:160-171)
            outputs = OrderedDict(res)
            m.assert_outputs(outputs)
            image = outputs.pop("_image")
            adjust = outputs.pop("adjust")        # zero_dce.py:168: the loss consumes it, the step output does not carry it
            outputs["loss"] = self.loss(image, adjust, outputs["enhanced"])
        else:
            outputs = OrderedDict(enhanced=res)
            m.assert_outputs(outputs)
            target = datapoint.get("ref_image")
            outputs["loss"] = self.loss(res, target.to(self.device, torch.float32))
        outputs["_tape"] = tape
        return outputs

    def backward(self, outputs: dict):
        tape, pred, loss = outputs["_tape"], outputs["enhanced"], outputs["loss"]
        if loss.seed_fn is not None:
            loss.seed_fn(tape, self.loss_scale) if self.loss_scale != 1.0 else loss.seed_fn(tape)
        else:
            g = torch.empty_like(pred, memory_format=CL)
            loss.backward_into(g, False, self.loss_scale)
            tape.seed(pred, g)
        self._pending, self._bucket_off = [], None
        tape.on_grad_bucket = self._early_allreduce if (self.overlap_allreduce and self.sync_gradients and self._world() > 1) else None
        if self.overlap_wgrad and self.compute_dtype != "f32":
            if self._wgrad_stream is None:
                # higher priority than the main stream: the weight-gradient kernel's few fat workgroups (144 accumulator VGPRs, LDS tiles) are placed first and the thin
                # workgroups of the streaming passes fill the wave slots beside them; at equal priority a streaming kernel's thousands of workgroups take every slot
                # and the two kernels run one after the other (IE_WGRAD_PRIO=0: equal priority, A/B)
                self._wgrad_stream = torch.cuda.Stream(device=self.device, priority=-1 if _os_env("IE_WGRAD_PRIO", "1") != "0" else 0)
            tape.wgrad_stream = self._wgrad_stream
        self._tape_in_backward = tape
        tape.backward()
        self._tape_in_backward = None

    def _early_allreduce(self, offset: int):
        import torch.distributed as dist
        if self._bucket_off is not None or offset <= 0 or offset >= self.store.numel:
            return                                # one early bucket per step
        self._bucket_off = offset
        if getattr(self, "_tape_in_backward", None) is not None:
            self._tape_in_backward.join_side()     # the tail's weight gradients may still be running on the side stream
        self._pending.append(dist.all_reduce(self.store.grads[offset:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def optimizer_step(self):
        """DDP mean of the gradients (one flat all-reduce over RCCL) -> clip_grad_norm_ -> Adam, three launches."""
        from .sharding import allreduce_gradients
        st = self.store
        if not self.sync_gradients:
            scale = 1.0 / self.loss_scale
        elif self._pending:                       # the tail went out during the backward pass: now the head, then wait for both
            import torch.distributed as dist
            self._pending.append(dist.all_reduce(st.grads[:self._bucket_off], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            for w in self._pending:
                w.wait()
            self._pending = []
            scale = 1.0 / self._world() / self.loss_scale
        else:
            scale = allreduce_gradients(st.grads, self.group) / self.loss_scale
        L = capi.lib()
        with torch.cuda.device(self.device):
            sumsq = None
            # the norm is needed for clipping AND as the overflow check of 16-bit / loss-scaled gradients: without it inf / NaN
            # gradients would be written into the fp32 master weights and both Adam moments for good
            if self.clip > 0 or self.compute_dtype == "f16" or self.loss_scale != 1.0:
                capi.check(L.ie_sumsq_f32(capi.ptr(st.grads), st.numel, scale, capi.ptr(self._sumsq), capi.ptr(self._sumsq_ws), 1,
                                          capi.current_stream()), "ie_sumsq_f32")
                sumsq = self._sumsq
            capi.check(L.ie_adam_step_counted_f32(capi.ptr(st.params), capi.ptr(st.grads), capi.ptr(st.exp_avg), capi.ptr(st.exp_avg_sq),
                                                  st.numel, capi.ptr(self._opt_state), self.lr, self.betas[0], self.betas[1], self.eps,
                                                  self.weight_decay, scale, capi.ptr(sumsq), self.clip, capi.current_stream()),
                       "ie_adam_step_counted_f32")
        st.repack16()                             # 16-bit copies of the weights for the next step's matrix-core convolutions

    def training_step(self, batch: dict, batch_idx: int = 0) -> dict:
        """nn/model.py:701-746 (+ the optimiser step Lightning runs right after it)."""
        outputs = self.forward_loss(batch)
        self.backward(outputs)
        self.optimizer_step()
        outputs.pop("_tape")
        return outputs

    def grad_norm(self) -> float:
        """Total gradient norm seen by the last optimizer_step (after the all-reduce mean)."""
        return float(self._sumsq.item()) ** 0.5

    def state_dict(self):
        return self.store.export()

    def gradients(self):
        """Checkpoint-layout gradients of the last backward pass (loss scale divided out)."""
        g = self.store.export(self.store.grads)
        if self.loss_scale != 1.0:
            for k in g:
                g[k] /= self.loss_scale
        return g

    # ---- checkpoint / resume (Lightning's ModelCheckpoint writes {"state_dict", "optimizer_states", "global_step", ...};
    #      nn/callback/model_checkpoint.py:35-185 -- the same top-level keys are used so mon's loaders find "state_dict") ----
    def save_checkpoint(self, path, weights_only: bool = False, epoch: int = 0) -> str:
        """Lightning-shaped checkpoint: ``state_dict`` / ``epoch`` / ``global_step`` always (what mon's loaders read,
        nn/model.py:45-76, 129-130), the optimiser state unless ``weights_only`` (the ``.pt`` twin of ModelCheckpoint)."""
        st = self.store
        ckpt = {"state_dict": self.state_dict(), "epoch": int(epoch), "global_step": self.step, "model": self.model.name}
        if not weights_only:
            ckpt["optimizer_states"] = [{"name": "adam", "lr": self.lr, "betas": list(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                                         "exp_avg": self.store.export(st.exp_avg), "exp_avg_sq": self.store.export(st.exp_avg_sq)}]
            ckpt["gradient_clip_val"] = self.clip
        import os
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        if not multi or dist.get_rank(self.group) == 0:       # data parallel: the replicas are bit-identical, rank 0 writes (Lightning's rank_zero_only)
            # No collective in here: a caller may guard the call with ``if rank == 0`` (a barrier would hang it until the collective times
            # out) or call it on every rank -- both work.  The file appears atomically (temporary name + os.replace), so a reader on another
            # rank sees either no file / the previous one or the complete new one, never a partial write.
            os.makedirs(os.path.dirname(os.path.abspath(str(path))) or ".", exist_ok=True)
            tmp = f"{path}.tmp.{os.getpid()}"
            try:
                torch.save(ckpt, tmp)
                os.replace(tmp, str(path))
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        return str(path)

    def load_checkpoint(self, path):
        """Resume: parameters, BatchNorm buffers, Adam moments and the step counter; the next step continues bit-identically."""
        ckpt = torch.load(str(path), map_location="cpu", weights_only=True)     # dicts / lists / numbers / strings / tensors only
        st = self.store
        if "state_dict" not in ckpt:              # a bare state_dict (the reference's weight files, nn/model.py:129-130)
            ckpt = {"state_dict": ckpt}
        if not ckpt.get("optimizer_states"):
            # weights-only file (ModelCheckpoint's .pt twin, save_weights_only): parameters and buffers are taken, the optimiser starts afresh
            # -- zero moments, step 0 -- instead of a KeyError; this trainer's own hyper-parameters stay
            opt = None
        else:
            opt = ckpt["optimizer_states"][0]

        def put(flat, tensors):
            for k, (o, n, shape, is_conv) in st.index.items():
                t = tensors[k].to(torch.float32)
                if tuple(t.shape) != tuple(shape):
                    raise RuntimeError(f"checkpoint entry {k} has shape {tuple(t.shape)}, expected {shape}")
                if is_conv:
                    ops.pack_hwio_into(t, flat[o:o + n])
                else:
                    flat[o:o + n].copy_(t.reshape(-1))
        put(st.params, ckpt["state_dict"])
        for k in st.buffers:
            st.buffers[k].copy_(ckpt["state_dict"][k])
        if opt is None:
            st.exp_avg.zero_(); st.exp_avg_sq.zero_()
            self.step = 0
        else:
            put(st.exp_avg, opt["exp_avg"])
            put(st.exp_avg_sq, opt["exp_avg_sq"])
            self.step = int(ckpt["global_step"])
            self.lr, self.betas, self.eps, self.weight_decay = float(opt["lr"]), tuple(opt["betas"]), float(opt["eps"]), float(opt["weight_decay"])
        st.repack16()                             # 16-bit graph: packed weight images made before the load are stale
        return self

    def sync_to_model(self):
        """Write the trained parameters back into the model object (for inference / checkpointing)."""
        sd = self.state_dict()
        for k, v in sd.items():
            self.model._params[k] = v
        self.model._on_params_change()
        return self.model


# ---- checkpoint callback with the reference's file naming (nn/callback/model_checkpoint.py:35-185) ---------------------------------
TORCH_FILE_FORMATS = (".pt", ".pth", ".weights", ".ckpt", ".tar")        # globals.py:708 (is_torch_file)


class ModelCheckpoint:
    """``<dirpath>/weights/<filename>_best[_<metric>].ckpt`` and, with ``save_last``, ``<filename>_last.ckpt``; every ``.ckpt`` gets a
    weights-only ``.pt`` twin (model_checkpoint.py:167-170), because mon's weight loaders accept ``.pt`` / ``.pth`` only
    (globals.py:709).  Always top-1, like the reference (``save_top_k`` forced to 1).  ``filename`` defaults to the model's
    ``fullname`` (:89-110).  Call ``on_validation_end(trainer, epoch, metrics)`` (or ``save``) from the fit loop."""

    CHECKPOINT_JOIN_CHAR, CHECKPOINT_NAME_LAST, CHECKPOINT_NAME_BEST, FILE_EXTENSION = "_", "last", "best", ".ckpt"

    def __init__(self, dirpath, filename: str | None = None, monitor: str | None = None, mode: str = "min", save_last: bool = False,
                 save_weights_only: bool = False, auto_insert_metric_name: bool = True):
        import os
        if mode not in ("min", "max"):
            raise ValueError(f"`mode` must be 'min' or 'max', but got {mode}.")
        dirpath = str(dirpath)
        self.dirpath = dirpath if os.path.basename(os.path.normpath(dirpath)) == "weights" else os.path.join(dirpath, "weights")
        self.filename, self.monitor, self.mode, self.save_last = filename, monitor, mode, save_last
        self.save_weights_only, self.auto_insert_metric_name = save_weights_only, auto_insert_metric_name
        self.best_model_score, self.best_model_path, self.last_model_path = None, "", ""
        self._last_epoch_saved = self._last_global_step_saved = 0

    def format_checkpoint_name(self, kind: str) -> str:
        name = self.CHECKPOINT_NAME_LAST if "last" in kind else self.CHECKPOINT_NAME_BEST
        if name == self.CHECKPOINT_NAME_BEST and self.auto_insert_metric_name and self.monitor:
            name += "_" + self.monitor.replace("/", "_")
        if self.filename not in (None, "None", ""):
            name = f"{self.filename}_{name}"
        import os
        return os.path.join(self.dirpath, name + self.FILE_EXTENSION)

    def _write(self, trainer, path, epoch):
        trainer.save_checkpoint(path.replace(self.FILE_EXTENSION, ".pt"), True, epoch)
        trainer.save_checkpoint(path, self.save_weights_only, epoch)
        self._last_epoch_saved, self._last_global_step_saved = int(epoch), int(trainer.step)

    def on_validation_end(self, trainer, epoch: int = 0, metrics: dict | None = None) -> list[str]:
        """Saves ``best`` when the monitored value improved (or always without a monitor) and ``last`` when asked to."""
        if self.filename is None and getattr(trainer.model, "fullname", None):
            self.filename = trainer.model.fullname
        written = []
        score = None if self.monitor is None else float((metrics or {})[self.monitor])
        better = score is None or self.best_model_score is None or (score < self.best_model_score if self.mode == "min" else score > self.best_model_score)
        if better:
            self.best_model_score, self.best_model_path = score, self.format_checkpoint_name("best")
            self._write(trainer, self.best_model_path, epoch)
            written.append(self.best_model_path)
        if self.save_last:
            self.last_model_path = self.format_checkpoint_name("last")
            self._write(trainer, self.last_model_path, epoch)
            written.append(self.last_model_path)
        return written

    save = on_validation_end

    def state_dict(self) -> dict:
        return {"dirpath": self.dirpath, "monitor": self.monitor, "last_epoch_saved": self._last_epoch_saved,
                "last_global_step_saved": self._last_global_step_saved, "best_model_score": self.best_model_score,
                "best_model_path": self.best_model_path, "last_model_path": self.last_model_path}


def get_latest_checkpoint(dirpath):
    """Newest torch file under ``dirpath`` by modification time (nn/model.py:79-92); None when there is none."""
    import os
    found = []
    for root, _, files in os.walk(str(dirpath)):
        found += [os.path.join(root, f) for f in files if f.endswith(TORCH_FILE_FORMATS)]
    return max(found, key=lambda f: os.stat(f).st_mtime) if found else None


def get_epoch_from_checkpoint(ckpt) -> int:
    return 0 if ckpt is None else int(torch.load(str(ckpt), map_location="cpu", weights_only=True).get("epoch", 0))


def get_global_step_from_checkpoint(ckpt) -> int:
    return 0 if ckpt is None else int(torch.load(str(ckpt), map_location="cpu", weights_only=True).get("global_step", 0))
