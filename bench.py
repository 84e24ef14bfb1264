#!/usr/bin/env python
"""Headline benchmark: enhanced images/sec @1920x1080, Zero-DCE, batch 64 per GPU (BASELINE.json
configs[1]), bf16 storage / fp32 accumulate, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (ZeroDCE_RE forward: NCHW fp32 frames in -> `enhanced` NCHW
fp32 out) over one batch of 64 synthetic frames per GPU.  Frames are independent, so ranks shard
the frames with no data-path collective ("scaling": "weak": per-GPU work is fixed).  Rank 0 prints
ONE JSON line; see DESIGN.md "Measurement" for the definition of every field.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 1080, 1920
BATCH_PER_GPU = 64
FLOP_PER_PX = 158400                      # 2 * 79,200 MAC
# per-kernel algorithmic work per pixel: (FLOP, HBM bytes with 16-bit activations and fp32 image/enhanced)
_L = [2 * 27 * 32, 2 * 288 * 32, 2 * 288 * 32, 2 * 288 * 32, 2 * 576 * 32, 2 * 576 * 32, 2 * 576 * 24]
KERNEL_WORK = {
    "zdce_conv1": (_L[0], 12 + 64), "zdce_conv2": (_L[1], 128), "zdce_conv3": (_L[2], 128), "zdce_conv4": (_L[3], 128),
    "zdce_conv5": (_L[4], 192), "zdce_conv6": (_L[5], 192), "zdce_conv7_curve": (_L[6], 128 + 12 + 12),
    # fused strip kernels: the intermediate of the pair stays in LDS
    "zdce_k12_conv1_conv2": (_L[0] + _L[1] + 2 * 288 * 24, 12 + 64 + 64),   # image in; x2 and P = W7a*x1 + b7 out
    "zdce_k45_conv4_conv5": (_L[3] + _L[4], 64 + 64),                # x3 in, x5 out (x4 in LDS)
    "zdce_k345_conv3_conv4_conv5": (_L[2] + _L[3] + _L[4], 64 + 64), # x2 in, x5 out (x3, x4 in LDS)
    "zdce_k345x_conv3_conv4_conv5": (_L[2] + _L[3] + _L[4], 64 + 64),        # round 4: the same layers with both cout fragments per wave (half the LDS fragment reads)
    "zdce_k67_conv6_conv7_curve": (_L[5] + 2 * 288 * 24, 3 * 64 + 12 + 12),  # (round-2 form) x2, x5, P, image in, enhanced out
    "zdce_k67n_conv6_conv7_curve": (_L[5] + 2 * 288 * 24, 3 * 64 + 12),      # x2, x5, P' (P + the fp32 pixel) in, enhanced out
    # round-4 schedule (no P): the last kernel recomputes x1 from the image -- that second e_conv1 is NOT algorithmic work and is not counted
    "zdce_k12x_conv1_conv2": (_L[0] + _L[1], 12 + 64),                              # image in, x2 out: 265 FLOP/B, under the balance point -> HBM-bound
    "zdce_k67x_conv1_conv6_conv7_curve": (_L[5] + _L[6], 12 + 64 + 64 + 12),        # image, x2, x5 in, enhanced out (x1, x6 in LDS)
}
PEAK_MFMA_TFLOPS = 2500.0                 # dense bf16/f16, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0                     # spec; 6.29 TB/s is what a float4 copy measures on this chip (same guide)
ACHIEVABLE_HBM_GBS = 6290.0
# the 3x3 stages with Cin >= 32 are bounded by the dense MFMA peak; only a stand-alone e_conv1 is HBM-bound
MFMA_BOUND = {"zdce_conv2", "zdce_conv3", "zdce_conv4", "zdce_conv5", "zdce_conv6", "zdce_conv7_curve",
              "zdce_k12_conv1_conv2", "zdce_k45_conv4_conv5", "zdce_k345_conv3_conv4_conv5", "zdce_k345x_conv3_conv4_conv5", "zdce_k67_conv6_conv7_curve",
              "zdce_k67n_conv6_conv7_curve", "zdce_k67x_conv1_conv6_conv7_curve"}
PRECISION = {
    "bf16": "16-bit storage, fp32 accumulate: x3/x4/x5 and the weights contracted with them bf16 (v_mfma_f32_16x16x32_bf16); "
            "x1/x2/x6 and their weights fp16 (v_mfma_f32_16x16x32_f16) -- DESIGN.md 4a",
    "bf16_pure": "every 16-bit tensor and weight bf16, fp32 accumulate",
    "f16": "every 16-bit tensor and weight fp16, fp32 accumulate",
}


# what `dtype` says for each --dtype: the arithmetic the path multiplies in (fp32 accumulate everywhere).  "bf16" selects the MIXED
# schedule of DESIGN.md 4a for Zero-DCE (x3 / x4 / x5 bf16; x1 / x2 / x6 / P and their weights fp16), so the label names both.
DTYPE_LABEL = {"bf16": "bf16+f16", "bf16_pure": "bf16", "f16": "f16"}
# the BASELINE.json configs reachable through `--workload` (N = frames per GPU per step, weak scaling)
WORKLOADS = {
    "zero_dce": dict(batch=BATCH_PER_GPU, dtype="bf16"),
    "retinexnet": dict(batch=32, dtype="f16", gflop_per_frame=1374.5, model=dict(name="retinexnet"),
                       what="retinexnet batch={n}/GPU 1920x1080 (BASELINE.json configs[3]: 256 frames over 8 GPUs)"),
    "unet32": dict(batch=32, dtype="f16", gflop_per_frame=634.5, model=dict(name="unet", base_channels=32, num_classes=3, enhance=True), oracle="unet",
                   what="unet32 (bilinear decoder, enhance head) batch={n}/GPU 1920x1080 (BASELINE.json configs[2])"),
    "unet64_train": dict(batch=8, dtype="bf16", gflop_per_frame=3 * 2530.0, model=dict(name="unet", base_channels=64, num_classes=3, enhance=True),
                         what="unet64 training step (L1+SSIM, Adam, clip 0.1) batch={n}/GPU 1920x1080 (BASELINE.json configs[4])"),
}


def _conv_census(torch, ops, fn):
    """HIP events around every matrix-core conv call of one pass of ``fn`` (tools/conv_census.py): {shape key: [calls, ms, flop]}."""
    import collections
    rec = collections.OrderedDict()
    orig = ops.mconv2d

    def logged(srcs, mw, stride=1, pad=1, **kw):
        s_ = [srcs] if isinstance(srcs, torch.Tensor) else list(srcs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(srcs, mw, stride=stride, pad=pad, **kw)
        e1.record()
        e1.synchronize()
        o = out[0] if isinstance(out, tuple) else out
        key = "conv{}x{} {}->{} s{} @{}x{}".format(mw.kh, mw.kw, "+".join(str(int(t.shape[1])) for t in s_), mw.cout, stride, int(s_[0].shape[2]), int(s_[0].shape[3]))
        r = rec.setdefault(key, [0, 0.0, 0.0])
        r[0] += 1
        r[1] += e0.elapsed_time(e1)
        r[2] += 2.0 * o.shape[0] * o.shape[2] * o.shape[3] * o.shape[1] * sum(int(t.shape[1]) for t in s_) * mw.kh * mw.kw
        return out

    orig_pair = ops.mconv2d_pair

    def logged_pair(x, mwa, mwb, pad_mode="zeros"):        # two stacked 3x3 layers in one launch (ie_mconv_pair_nhwc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_pair(x, mwa, mwb, pad_mode=pad_mode)
        e1.record()
        e1.synchronize()
        key = "pair of conv3x3 {}->{}->{} @{}x{} (ie_mconv_pair_nhwc)".format(int(x.shape[1]), mwa.cout, mwb.cout, int(x.shape[2]), int(x.shape[3]))
        r = rec.setdefault(key, [0, 0.0, 0.0])
        r[0] += 1
        r[1] += e0.elapsed_time(e1)
        r[2] += 2.0 * out.shape[0] * out.shape[2] * out.shape[3] * 9 * (int(x.shape[1]) * mwa.cout + mwa.cout * mwb.cout)
        return out

    def timed(key, flop, call):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = call()
        e1.record()
        e1.synchronize()
        r = rec.setdefault(key, [0, 0.0, 0.0])
        r[0] += 1
        r[1] += e0.elapsed_time(e1)
        r[2] += flop
        return out

    orig_thin, orig_k, orig_up = ops.mconv2d_pair_thin, ops.conv9x9_c4, ops.unet_up_head

    def logged_thin(x, mwa, mwb, **kw):                     # Conv3x3(C, C) + ReLU and a thin Conv3x3 behind it (ie_mconv_pair_thin_nhwc)
        px = float(x.shape[0] * x.shape[2] * x.shape[3])
        return timed("pair of conv3x3 {}->{}->{} @{}x{} (ie_mconv_pair_thin_nhwc)".format(int(x.shape[1]), mwa.cout, mwb.cout, int(x.shape[2]), int(x.shape[3])),
                     2.0 * px * 9 * (int(x.shape[1]) * mwa.cout + mwa.cout * mwb.cout), lambda: orig_thin(x, mwa, mwb, **kw))

    def logged_k(x4, cw):                                   # first layer on a 4-channel image (ie_conv9x9_c4_nhwc / ie_conv3x3_c4_nhwc)
        px = float(x4.shape[0] * x4.shape[2] * x4.shape[3])
        return timed("conv{0}x{0} 4->64 @{1}x{2} (ie_conv{0}x{0}_c4_nhwc)".format(cw.k, int(x4.shape[2]), int(x4.shape[3])), 2.0 * px * 64 * 4 * cw.k * cw.k,
                     lambda: orig_k(x4, cw))

    def logged_up(skip, up, mwa, mwb, tail, clamp=None):    # U-Net's last decoder level + head (ie_unet_up_head_nhwc)
        px, c = float(skip.shape[0] * skip.shape[2] * skip.shape[3]), int(skip.shape[1])
        return timed("up-sample + cat + DoubleConv({}->{}->{}) + 1x1 head @{}x{} (ie_unet_up_head_nhwc)".format(2 * c, c, c, int(skip.shape[2]), int(skip.shape[3])),
                     2.0 * px * (9 * (2 * c * c + c * c) + c * tail.cout), lambda: orig_up(skip, up, mwa, mwb, tail, clamp=clamp))

    ops.mconv2d, ops.mconv2d_pair, ops.mconv2d_pair_thin, ops.conv9x9_c4, ops.unet_up_head = logged, logged_pair, logged_thin, logged_k, logged_up
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        ops.mconv2d, ops.mconv2d_pair, ops.mconv2d_pair_thin, ops.conv9x9_c4, ops.unet_up_head = orig, orig_pair, orig_thin, orig_k, orig_up
    return rec


def run_generator_workload(args, ctx):
    """`--workload unet32 | retinexnet | unet64_train`: the same contract as the headline (W untimed + K timed steps between barriers,
    max over ranks, whole-job frames/s), one JSON line on rank 0."""
    torch, mon_amd, dist, dev, rank, world = ctx["torch"], ctx["mon_amd"], ctx["dist"], ctx["dev"], ctx["rank"], ctx["world"]
    from mon_amd import ops
    wl = WORKLOADS[args.workload]
    n = args.batch
    train = args.workload == "unet64_train"
    model = mon_amd.MODELS.build(seed=1, compute_dtype="f32" if train else args.dtype, **wl["model"]).to(dev)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    if train:
        tr = mon_amd.Trainer(model, optimizer={"name": "adam", "lr": 1e-4}, loss={"name": "l1_ssim_loss"}, gradient_clip_val=0.1, compute_dtype=args.dtype)
        tgt = torch.rand((n, 3, H, W), generator=g, device=dev)
        img = (tgt ** 2.2 * 0.4).contiguous()
        step = lambda: tr.training_step({"image": img, "ref_image": tgt})      # noqa: E731
    else:
        x = torch.rand((n, 3, H, W), generator=g, device=dev)
        step = lambda: model.forward({"image": x})                              # noqa: E731

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(k):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        sync_all()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=ctx["red_dev"])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        step()
    dt = timed(args.steps)
    ms_per_step = dt / args.steps * 1e3
    value = n * world * args.steps / dt
    comm = None
    if train and world > 1:
        # the exchange step by itself, and the step without it / without the overlap (same K, same barriers)
        tr.sync_gradients = False
        ms_no = timed(args.steps) / args.steps * 1e3
        tr.sync_gradients, tr.overlap_allreduce = True, False
        ms_serial = timed(args.steps) / args.steps * 1e3
        tr.overlap_allreduce = True
        sync_all()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dist.all_reduce(tr.store.grads)
        e1.record()
        torch.cuda.synchronize(dev)
        comm = {"allreduce_ms": round(e0.elapsed_time(e1) / 5, 3), "allreduce_bytes": int(tr.store.grads.numel() * 4),
                "ms_per_step_without_allreduce": round(ms_no, 3), "ms_per_step_allreduce_after_backward": round(ms_serial, 3),
                "ms_per_step_overlapped": round(ms_per_step, 3), "backend": dist.get_backend()}
    if rank != 0:
        return None
    # ---- roofline of the dominant conv shape: HIP events around every matrix-core conv call of one more pass ------------------
    peak_frac = value / world * wl["gflop_per_frame"] * 1e9 / (PEAK_MFMA_TFLOPS * 1e12)
    roofline = {"kernel": None, "bound": "mfma", "achieved": round(value / world * wl["gflop_per_frame"] / 1e3, 1), "peak": PEAK_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(peak_frac, 4), "traffic": None, "model_mfma_frac": round(peak_frac, 4),
                "note": "whole-step algorithmic FLOPs (SURVEY 8d) / step time"}
    try:
        if train:
            tr.sync_gradients = False          # rank 0 alone runs this extra pass: no collective in it
        rec = _conv_census(torch, ops, step)
        key, (calls, ms, flop) = max(rec.items(), key=lambda kv: kv[1][1])
        tot = sum(r[1] for r in rec.values())
        ach = flop / (ms * 1e-3) / 1e12
        roofline.update({"kernel": key if "(ie_" in key else "ie_mconv_nhwc " + key, "achieved": round(ach, 1), "frac": round(ach / PEAK_MFMA_TFLOPS, 4),
                         "avg_launch_ms": round(ms / calls, 4), "launches_per_step": calls, "share_of_conv_time": round(ms / tot, 3),
                         "conv_ms_per_step": round(tot, 2),
                         "note": "dominant conv shape of the step: algorithmic FLOPs (2*MAC) / HIP-event time around its ie_mconv_nhwc calls"
                                 + (" (forward and data-gradient calls; the weight-gradient kernel is outside this census)" if train else "")})
    except Exception as e:       # noqa: BLE001
        roofline["census_error"] = str(e)[:200]
    try:      # HBM bytes per launch of that kernel from the committed PMC passes (profiles/traffic.json "workloads"), when it is the kernel they were taken on
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tw = json.load(f).get("workloads", {}).get(args.workload)
        if tw and roofline.get("kernel") and tw["kernel_contains"] in roofline["kernel"] and n == tw.get("batch", 32) and (H, W) == (1080, 1920):
            roofline.update({"traffic": tw["traffic"], "traffic_source": tw["source"], "mfma_busy_frac_pmc": tw.get("mfma_busy_frac"), "clock_GHz_pmc": tw.get("clock_GHz")})
    except Exception:       # noqa: BLE001
        pass
    cpu_baseline = parity = None
    if world == 1 and not args.no_cpu_baseline:
        ncpu = _usable_cpus()
        torch.set_num_threads(max(1, ncpu))
        from oracle import generators as o_gen
        if train:
            # a bounded sample of the same step on the host: forward + autograd backward of the oracle graph on a 270x480 crop
            # (1/16 of a frame), scaled by area -- the reference has no CPU trainer for this config, so this is a "port"
            from oracle import train as o_train
            sd0 = {k: v.clone() for k, v in model.state_dict().items()}
            xs_, ts_ = img[:1, :, :270, :480].cpu(), tgt[:1, :, :270, :480].cpu()
            t0 = time.perf_counter()
            o_train.training_steps(sd0, [(xs_, ts_)], lambda p_, b_, x_: o_train.unet_forward_train(p_, b_, x_, "bilinear", True),
                                   lambda a_, b_: o_train.l1_ssim_loss(a_, b_), lr=1e-4, clip=0.1)
            sec = (time.perf_counter() - t0) * 16
            cpu_baseline = {"value": round(1.0 / sec, 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": _cpu_model(),
                            "sample": "one optimiser step of oracle/train.py (torch autograd, fp32) on one 270x480 crop, time x 16"}
        else:
            xs_ = x[:1].cpu()
            sd_ = {k: v.clone() for k, v in model.state_dict().items()}
            if wl.get("oracle") == "unet":
                o_fwd, o_name = (lambda sd__, t_: {"enhanced": o_gen.unet_forward(sd__, t_, enhance=True)["enhanced"]}), "unet_forward"
            else:
                o_fwd, o_name = o_gen.retinexnet_forward, "retinexnet_forward"
            with torch.no_grad():
                o_fwd(sd_, xs_[:, :, :64, :64])
                t0 = time.perf_counter()
                ref = o_fwd(sd_, xs_)["enhanced"]
                sec = time.perf_counter() - t0
            got = model.forward({"image": x[:1]})["enhanced"].cpu()
            cpu_baseline = {"value": round(1.0 / sec, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": _cpu_model(),
                            "sample": f"one 1920x1080 frame, torch CPU restatement (oracle/generators.py {o_name}), one run after a 64x64 warm-up"}
            parity = {"max_rel_err": float(f"{float((got - ref).abs().max() / ref.abs().max()):.3e}")}
    line = {"metric": "train_images_per_sec_1080p" if train else "enhanced_images_per_sec_1080p", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": wl["what"].format(n=n), "frames_per_gpu": n, "global_batch": n * world, "height": H, "width": W,
                       "precision": f"{args.dtype} storage of activations" + (" and their gradients; fp32 master weights, statistics, weight gradients, Adam" if train else "; fp32 accumulate"),
                       "parallelism": (f"data parallel over {world} GPU(s): one flat-buffer gradient all-reduce per step, decoder half overlapped with the encoder's backward"
                                       if train else f"frames sharded over {world} GPU(s), no collective"),
                       "tflops_per_gpu": round(value / world * wl["gflop_per_frame"] / 1e3, 1)},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity}
    if comm is not None:
        line["comm"] = comm
    return line


def _spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run, one process per
    GPU) BEFORE this process touches the GPU, and hand their exit code back.  Rank 0 of the children prints the JSON line."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(_spawn_cmd(n, sys.argv[1:]), env=env)


def _spawn_cmd(n, argv):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"



def _usable_cpus() -> int:
    """Host cores this process may actually use: the affinity mask, cut to the cgroup's CPU quota (a 1-GPU box of the pool shows 256 CPUs in the mask and a quota of 16:
    256 worker threads on 16 cores' worth of time ran the CPU baseline five times slower than 16 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:       # noqa: BLE001
        pass
    return max(1, n)


def _parse_rocm_smi(text: str):
    """{socket_power_W, power_cap_W, sclk_MHz} out of `rocm-smi --showpower --showclocks --showmaxpower` text (None when power or sclk is missing)."""
    import re
    pw = re.search(r"Current Socket Graphics Package Power \(W\): ([\d.]+)", text) or re.search(r"Average Graphics Package Power \(W\): ([\d.]+)", text)
    cap = re.search(r"Max Graphics Package Power \(W\): ([\d.]+)", text)
    sclk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", text)
    if not (pw and sclk):
        return None
    return {"socket_power_W": float(pw.group(1)), "power_cap_W": float(cap.group(1)) if cap else None, "sclk_MHz": int(sclk.group(1))}


def _board_state(enqueue_step, drain, step_ms: float, device_index: int = 0):
    """Socket power, power cap and shader clock WHILE the workload runs: ~1.5 s of extra, untimed steps are enqueued, a read-only `rocm-smi` poll runs beside them
    0.6 s in, then the stream is drained (`drain`).  The Zero-DCE schedule sits at the board's power cap (finding 16): the clock the chip holds there, not the
    pipe cycles a kernel needs, sets its MFMA rate -- the line a reader needs next to `roofline.frac`.  None when rocm-smi is absent or fails; never raises."""
    import shutil
    import subprocess
    import time
    try:
        if shutil.which("rocm-smi") is None:
            return None
        k = max(20, min(400, int(1500.0 / max(step_ms, 0.1))))
        for _ in range(k):
            enqueue_step()
        time.sleep(0.6)
        out = subprocess.run(["rocm-smi", "-d", str(device_index), "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=20).stdout
        drain()
        st = _parse_rocm_smi(out)
        if st is None:
            return None
        st["how"] = f"rocm-smi polled 0.6 s into {k} extra untimed steps"
        return st
    except Exception as e:       # noqa: BLE001
        print(f"[bench] board probe skipped: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        try:
            drain()
        except Exception:   # noqa: BLE001
            pass
        return None

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="zero_dce", choices=sorted(WORKLOADS),
                    help="zero_dce = BASELINE.json configs[1] (the headline, default); retinexnet = configs[3] (32 frames per GPU, frames "
                         "sharded over the ranks, no collective); unet64_train = configs[4] (U-Net-64, L1+SSIM, 8 frames per GPU, one RCCL "
                         "all-reduce of the flat gradient buffer per step, its tail overlapped with the encoder's backward pass)")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default: the workload's BASELINE.json batch)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "bf16_pure", "f16"], help="default: bf16 (zero_dce: the mixed fp16 / bf16 schedule), f16 for retinexnet")
    ap.add_argument("--output", default="f32", choices=["f32", "u8"], help="enhanced as fp32 NCHW or uint8 NHWC")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant bitmask (capi.VAR_*), A-B runs only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-board", action="store_true", help="skip the rocm-smi poll (socket power / sclk while the workload runs)")
    ap.add_argument("--no-extras", action="store_true", help="skip the short side measurements of the other BASELINE configs")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    args.batch = args.batch or wl["batch"]
    args.dtype = args.dtype or wl["dtype"]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(args.gpus))      # no GPU call has happened in this process

    import torch
    import mon_amd
    from mon_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    # IE_BENCH_BACKEND=gloo + IE_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank control flow on a 1-GPU box
    backend = os.environ.get("IE_BENCH_BACKEND", "nccl")
    if os.environ.get("IE_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            except TypeError:
                dist.init_process_group("nccl", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus and dist.get_backend() == backend
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    if args.workload != "zero_dce":
        ctx = dict(rank=rank, world=world, dev=dev, dist=dist, red_dev=red_dev, torch=torch, mon_amd=mon_amd, capi=capi)
        line = run_generator_workload(args, ctx)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from oracle import zero_dce as o_zdce      # only for the seeded weight rule + the CPU baseline leg
    sd = o_zdce.init_state_dict(seed=1, gain=3.0)
    model = mon_amd.MODELS.build(name="zero_dce_re", weights=dict(sd), compute_dtype=args.dtype, variant=args.variant).to(dev).eval()

    n = args.batch
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.rand((n, 3, H, W), generator=g, device=dev, dtype=torch.float32)
    out = {"enhanced": torch.empty_like(x)} if args.output == "f32" else {"u8": torch.empty((n, H, W, 3), device=dev, dtype=torch.uint8)}
    kw = dict(want_enhanced=args.output == "f32", want_u8=args.output == "u8", out=out)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        model.enhance(x, **kw)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.enhance(x, **kw)
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    value = n * world * args.steps / dt

    roofline = None
    cpu_baseline = None
    parity = None
    if rank == 0:
        # ---- per-kernel timing with hipEvents on the launch stream (ie_zdce_profile_forward) ----
        import ctypes as C
        L = capi.lib()
        plan = model._ensure_plan()
        ms = (C.c_float * 8)()
        names = (C.c_char_p * 8)()
        launches = (C.c_int * 8)()
        cnt = C.c_int(0)
        reps = 3
        acc = [0.0] * 7
        nl = [0] * 7
        for _ in range(reps):
            capi.check(L.ie_zdce_profile_forward(plan, capi.ptr(x), capi.ptr(out.get("enhanced")), None, capi.ptr(out.get("u8")),
                                                 n, H, W, capi.current_stream(), ms, names, launches, 8, C.byref(cnt)))
            for i in range(7):
                acc[i] += ms[i] / reps
                nl[i] = launches[i]
        layer_ms = {names[i].decode(): round(acc[i], 4) for i in range(7) if names[i]}
        dom = max(range(7), key=lambda i: acc[i])
        kflop, kbytes = KERNEL_WORK[names[dom].decode()]
        px_per_launch = n * H * W / max(nl[dom], 1)
        avg_ms = acc[dom] / max(nl[dom], 1)
        flops = kflop * px_per_launch
        byts = kbytes * px_per_launch
        kname = names[dom].decode()
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        tsrc = None
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic, tsrc = tj.get(kname), tj.get("_source")
                ref_px = tj.get("_px_per_launch", 16 * 1080 * 1920)         # the PMC passes were taken on 16-frame 1080p launches
                if traffic is not None and int(px_per_launch) != ref_px:    # another launch size: scale by pixels and say so
                    traffic = int(traffic * px_per_launch / ref_px)
                    tsrc = f"{tsrc}; scaled by pixels per launch ({int(px_per_launch)} / {ref_px})"
            except Exception:
                traffic = None
        ach_tflops = flops / (avg_ms * 1e-3) / 1e12
        ach_gbs = byts / (avg_ms * 1e-3) / 1e9
        if kname in MFMA_BOUND:
            # 3x3 stages with Cin >= 32 -> dense-MFMA roofline.  achieved = algorithmic FLOPs of the layers the
            # kernel fuses (2*MAC, unpadded) x pixels per launch / measured launch time.  The bytes the fused SCHEDULE moves
            # (x2 / x5 / P re-reads included) are reported beside it against the HBM peak, as secondary information.
            roofline = {"kernel": kname, "bound": "mfma", "achieved": round(ach_tflops, 1), "peak": PEAK_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach_tflops / PEAK_MFMA_TFLOPS, 4), "traffic": traffic,
                        "traffic_source": tsrc, "avg_launch_ms": round(avg_ms, 4), "launches_per_step": nl[dom],
                        "flop_per_px": kflop, "px_per_launch": int(px_per_launch),
                        "hbm_schedule_bytes_per_px": kbytes, "hbm_schedule_GBps": round(ach_gbs, 1),
                        "hbm_schedule_frac_of_spec_8000": round(ach_gbs / PEAK_HBM_GBS, 4),
                        "hbm_schedule_frac_of_achievable_6290": round(ach_gbs / ACHIEVABLE_HBM_GBS, 4)}
        else:
            roofline = {"kernel": kname, "bound": "hbm", "achieved": round(ach_gbs, 1), "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": round(ach_gbs / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                        "avg_launch_ms": round(avg_ms, 4), "launches_per_step": nl[dom], "achievable_GBps": ACHIEVABLE_HBM_GBS}
        roofline["kernel_ms_per_step"] = layer_ms
        roofline["model_bytes_per_px"] = sum(KERNEL_WORK[k][1] for k in layer_ms)
        roofline["kernel_mfma_tflops"] = {k: round(KERNEL_WORK[k][0] * n * H * W / (v * 1e-3) / 1e12, 1) for k, v in layer_ms.items() if v > 0}
        roofline["model_mfma_frac"] = round(value / world * H * W * FLOP_PER_PX / (PEAK_MFMA_TFLOPS * 1e12), 4)
        if world == 1 and not args.no_board:
            roofline["board"] = _board_state(lambda: model.enhance(x, **kw), torch.cuda.synchronize, ms_per_step, torch.cuda.current_device())

        # The GPU side of the parity check runs here; the CPU legs (oracle timing + parity) run AFTER the side measurements: the
        # 64 CPU worker threads of the oracle slowed the launch-bound U-Net forward that followed them by a quarter.
        ref = xs = x4 = got_e = got_u8 = None
        pending16 = {}
        if world == 1 and not args.no_cpu_baseline:
            xs, x4 = x[:1].cpu(), x[:4].cpu()
            got = model.enhance(x[:1], want_enhanced=True, want_u8=True)
            got_e, got_u8 = got["enhanced"].cpu(), got["u8"].cpu()
            del got

        # ---- side measurements (N=1 only, a few seconds): the other BASELINE.json configs, NOT part of `value` ----------------
        extras = None
        if world == 1 and not args.no_extras:
            extras = {}
            del x
            torch.cuda.empty_cache()

            def timed(fn, reps):
                # median of individually synchronised repetitions (each >= 20 ms: the sync costs nothing measurable): one repetition that runs into an
                # allocator release of multi-GiB blocks (seen once in ~10 runs: 1,236 instead of 1,49x frames/s) must not colour a side measurement
                fn(); fn(); torch.cuda.synchronize()
                ts = []
                for _ in range(reps + 2):
                    t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                ts.sort()
                return ts[len(ts) // 2]

            # the same workload in the other two 16-bit modes: all-fp16 (the mode that holds 1e-3 on every golden) and all-bf16
            for od in ("f16", "bf16_pure"):
                try:
                    m16 = mon_amd.MODELS.build(name="zero_dce_re", weights=dict(sd), compute_dtype=od).to(dev).eval()
                    x16 = torch.rand(n, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
                    t16 = timed(lambda: m16.enhance(x16, want_enhanced=True), 3)
                    extras[f"zero_dce_{od}_1080p_frames_per_s"] = round(n / t16, 1)
                    if xs is not None:
                        pending16[od] = m16.enhance(xs.to(dev), want_enhanced=True)["enhanced"].cpu()
                    del m16, x16
                except Exception as e:
                    extras[f"zero_dce_{od}_1080p_frames_per_s"] = f"failed: {e}"[:200]
                torch.cuda.empty_cache()
            # BASELINE.json configs[2] (U-Net-32, batch 32) and configs[3] (RetinexNet, 32 frames per GPU) at their stated batch sizes, FIRST in the
            # configuration that is on the matrix cores AND inside the 1e-3 bound (fp16 storage, v_mfma_f32_16x16x32_f16): throughput at batch 32 and
            # the error of one 1080p frame against the CPU fp32 oracle
            if xs is not None:
                from oracle import generators as o_gen
                for tag, cfg, ref_fn in (("unet32_f16", dict(name="unet", base_channels=32, num_classes=3, enhance=True),
                                          lambda sd_, t: o_gen.unet_forward(sd_, t, "bilinear", True)["enhanced"]),
                                         ("retinexnet_f16", dict(name="retinexnet"), lambda sd_, t: o_gen.retinexnet_forward(sd_, t)["enhanced"])):
                    try:
                        m5 = mon_amd.MODELS.build(seed=1, compute_dtype="f16", **cfg).to(dev)
                        xi = torch.rand(32, 3, H, W, device=dev)
                        extras[f"{tag}_1080p_batch32_frames_per_s"] = round(32 / timed(lambda: m5.forward({"image": xi}), 3), 1)
                        del xi
                        pending16[tag] = (m5.forward({"image": xs.to(dev)})["enhanced"].cpu(), {k: v.clone() for k, v in m5.state_dict().items()}, ref_fn)
                        del m5
                    except Exception as e:
                        extras[f"{tag}_1080p_batch32_frames_per_s"] = f"failed: {e}"[:200]
                    torch.cuda.empty_cache()
            # ... then the same models with bf16 storage (same kernels and rate; 2e-3 .. 4e-3 of the fp32 result: outside the bound, for reference)
            for tag, cfg, nfr in (("unet32_bf16_1080p_batch32_frames_per_s", dict(name="unet", base_channels=32, num_classes=3, enhance=True), 32),
                                  ("unet32_pixelshuffle_bf16_1080p_batch32_frames_per_s", dict(name="unet", base_channels=32, num_classes=3, enhance=True, upsampler="pixelshuffle"), 32),
                                  ("retinexnet_bf16_1080p_batch32_frames_per_s", dict(name="retinexnet"), 32)):
                try:
                    m2 = mon_amd.MODELS.build(seed=1, compute_dtype="bf16", **cfg).to(dev)
                    xi = torch.rand(nfr, 3, H, W, device=dev)
                    extras[tag] = round(nfr / timed(lambda: m2.forward({"image": xi}), 3), 1)
                    del m2, xi
                except Exception as e:                       # a side measurement must never take the headline line down
                    extras[tag] = f"failed: {e}"[:200]
                torch.cuda.empty_cache()
            # BASELINE.json configs[4]'s step (U-Net-64, L1+SSIM, Adam + clip) at its 8 frames per GPU: the 16-bit graph, and the
            # fp32 parity graph at 2 frames
            for tag, cdt, nb in (("unet64_train_l1ssim_fp32_1080p_batch2_frames_per_s", "f32", 2), ("unet64_train_l1ssim_bf16_1080p_batch8_frames_per_s", "bf16", 8)):
                try:
                    m3 = mon_amd.MODELS.build(name="unet", base_channels=64, num_classes=3, enhance=True, seed=1).to(dev)
                    tr = mon_amd.Trainer(m3, optimizer={"name": "adam", "lr": 1e-4}, loss={"name": "l1_ssim_loss"}, gradient_clip_val=0.1,
                                         compute_dtype=cdt)
                    tgt = torch.rand(nb, 3, H, W, device=dev)
                    img = (tgt ** 2.2 * 0.4).contiguous()
                    extras[tag] = round(nb / timed(lambda: tr.training_step({"image": img, "ref_image": tgt}), 2), 2)
                    del m3, tr, tgt, img
                except Exception as e:
                    extras[tag] = f"failed: {e}"[:200]
                torch.cuda.empty_cache()

            try:      # zero-reference training of the headline model itself, 16-bit graph, reference optimiser config
                m4 = mon_amd.MODELS.build(name="zero_dce_re", seed=1).to(dev)
                tr4 = mon_amd.Trainer(m4, optimizer={"name": "adam", "lr": 5e-5, "betas": (0.9, 0.99), "weight_decay": 1e-5}, gradient_clip_val=0.1,
                                      compute_dtype="bf16")
                im4 = (torch.rand(8, 3, H, W, device=dev) ** 2.2 * 0.4).contiguous()
                extras["zero_dce_train_zeroref_bf16_1080p_frames_per_s"] = round(8 / timed(lambda: tr4.training_step({"image": im4}), 2), 1)
                del m4, tr4, im4
            except Exception as e:
                extras["zero_dce_train_zeroref_bf16_1080p_frames_per_s"] = f"failed: {e}"[:200]
            torch.cuda.empty_cache()

        # ---- CPU baseline (oracle, "port") + parity of this very configuration on one frame ----------------------------------
        if world == 1 and not args.no_cpu_baseline:
            ncpu = _usable_cpus()
            torch.set_num_threads(max(1, ncpu))             # every host core this process may use (affinity mask cut to the cgroup quota); `cores` reports the number
            def cpu_rate(batch):
                with torch.no_grad():
                    o_zdce.forward(sd, batch)               # warm-up
                    ts = []
                    out = None
                    for _ in range(3):
                        c0 = time.perf_counter()
                        out = o_zdce.forward(sd, batch)["enhanced"]
                        ts.append(time.perf_counter() - c0)
                ts.sort()
                return batch.shape[0] / ts[1], out
            r1, ref = cpu_rate(xs)
            r4, _ = cpu_rate(x4)
            cpu_baseline = {"value": round(r1, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                            "cpu_model": _cpu_model(), "os_cpu_count": os.cpu_count(), "value_batch4": round(r4, 4),
                            "sample": "1 frame (value) and 4 frames (value_batch4) of 1920x1080 fp32, torch CPU restatement "
                                      "(oracle/zero_dce.py), 1 warm-up + median of 3 each"}
            e = got_e
            err = float((e - ref).abs().max() / ref.abs().max())
            mse = float(((e.double() - ref.double()) ** 2).mean())
            import math
            u8ref = o_zdce.to_uint8_save_image(ref)
            d = (got_u8.int() - u8ref.int()).abs()
            parity = {"max_rel_err": float(f"{err:.3e}"), "psnr_db_vs_cpu_fp32": round(10 * math.log10(1.0 / mse), 2) if mse > 0 else None,
                      "u8_exact_match": round(float((d == 0).float().mean()), 5), "u8_max_abs_diff": int(d.max())}
            for od, e16 in pending16.items():
                if extras is None:
                    continue
                if isinstance(e16, tuple):                 # (engine output, weights, oracle forward) of a generator in fp16
                    got16, sd16, fn16 = e16
                    with torch.no_grad():
                        r16 = fn16(sd16, xs)
                    extras[f"{od}_max_rel_err"] = float(f"{float((got16 - r16).abs().max() / r16.abs().max()):.3e}")
                else:
                    extras[f"zero_dce_{od}_max_rel_err"] = float(f"{float((e16 - ref).abs().max() / ref.abs().max()):.3e}")

        line = {
            "metric": "enhanced_images_per_sec_1080p", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_LABEL[args.dtype], "data": "synthetic",
            "config": {"workload": f"zero_dce_re batch={n}/GPU {W}x{H} (BASELINE.json configs[1])", "frames_per_gpu": n,
                       "global_batch": n * world, "height": H, "width": W, "precision": PRECISION[args.dtype], "input": "NCHW fp32 in HBM",
                       "output": "enhanced NCHW fp32" if args.output == "f32" else "enhanced NHWC uint8",
                       "parallelism": f"frames sharded over {world} GPU(s), no collective", "tflops_per_gpu": round(value / world * H * W * FLOP_PER_PX / 1e12, 1)},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity,
        }
        if extras is not None:
            line["extras"] = extras
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
