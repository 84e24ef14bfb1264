#!/usr/bin/env python
"""Training-step throughput (BASELINE.json configs[4]: U-Net-64, L1+SSIM, batch 8/GPU at 1080p) on the fp32 path.
Usage: python tools/bench_train.py [--base 64] [--batch 8] [--hw 1080 1920] [--steps 3]
Multi-GPU: torchrun --nproc-per-node N tools/bench_train.py ...  (one flat-buffer all-reduce per step over RCCL).
Prints one JSON line per run (rank 0).  Not the headline metric (bench.py)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import mon_amd

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="unet", choices=["unet", "zero_dce", "llunetpp", "hinet"])
ap.add_argument("--base", type=int, default=64)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--hw", type=int, nargs=2, default=[1080, 1920])
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "bf16"], help="activation dtype of the training graph")
a = ap.parse_args()
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
if world > 1:
    dist.init_process_group("nccl")
if a.model == "unet":
    model = mon_amd.MODELS.build(name="unet", in_channels=3, num_classes=3, base_channels=a.base, enhance=True, seed=1).to(f"cuda:{local}")
    tr = mon_amd.Trainer(model, optimizer={"name": "adam", "lr": 1e-4}, loss={"name": "l1_ssim_loss"}, gradient_clip_val=0.1, compute_dtype=a.dtype)
elif a.model == "hinet":
    model = mon_amd.MODELS.build(name="hinet_re", seed=1).to(f"cuda:{local}")
    tr = mon_amd.Trainer(model, optimizer={"name": "adam", "lr": 1e-4}, loss={"name": "l1_ssim_loss"}, gradient_clip_val=0.1, compute_dtype=a.dtype)
elif a.model == "llunetpp":
    model = mon_amd.MODELS.build(name="llunet++_re", seed=1).to(f"cuda:{local}")
    tr = mon_amd.Trainer(model, optimizer={"name": "adam", "lr": 1e-4}, loss={"name": "l1_ssim_loss"}, gradient_clip_val=0.1, compute_dtype=a.dtype)
else:       # zero-reference training of Zero-DCE with the reference's optimiser config (zero_dce_re_sice_mix.py:46-51,106)
    model = mon_amd.MODELS.build(name="zero_dce_re", seed=1).to(f"cuda:{local}")
    tr = mon_amd.Trainer(model, optimizer={"name": "adam", "lr": 5e-5, "betas": (0.9, 0.99), "weight_decay": 1e-5}, gradient_clip_val=0.1,
                         compute_dtype=a.dtype)
g = torch.Generator().manual_seed(2 + rank)
tgt = torch.rand(a.batch, 3, *a.hw, generator=g).cuda()
img = (tgt ** 2.2 * 0.4).contiguous()
losses = []
for i in range(a.warmup + a.steps):
    if i == a.warmup:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
    out = tr.training_step({"image": img, "ref_image": tgt})
    losses.append(float(out["loss"]))
torch.cuda.synchronize()
dt = torch.tensor([(time.perf_counter() - t0) / a.steps], dtype=torch.float64, device="cuda")
if world > 1:
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
fwd_gflop = {64: 2530.0, 32: 634.5}.get(a.base)
if rank == 0:
    wl = {"unet": f"unet{a.base} train L1+SSIM", "llunetpp": "llunet++ train L1+SSIM", "hinet": "hinet-64 train L1+SSIM (two stages)", "zero_dce": "zero_dce train zero-reference loss"}[a.model]
    rec = {"workload": f"{wl} batch={a.batch}/GPU {a.hw[0]}x{a.hw[1]} {a.dtype}", "n_gpus": world, "s_per_step": round(dt.item(), 4),
           "images_per_s": round(a.batch * world / dt.item(), 3), "loss_first_last": [round(losses[0], 5), round(losses[-1], 5)],
           "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    if a.model == "unet" and fwd_gflop and tuple(a.hw) == (1080, 1920):
        rec["tflops_3x_fwd"] = round(3 * fwd_gflop * a.batch / dt.item() / 1e3, 1)
    print(json.dumps(rec), flush=True)
if world > 1:
    dist.destroy_process_group()
