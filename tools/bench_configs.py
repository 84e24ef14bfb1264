#!/usr/bin/env python
"""BASELINE.json configs[2] and configs[3] at their stated batch sizes (frames/s, peak memory).
One GPU:   python tools/bench_configs.py
N GPUs:    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_configs.py
           (frames are sharded over the ranks with no data-path collective; RCCL only carries the timing reduction)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import mon_amd

world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
if world > 1:
    dist.init_process_group("nccl")
dev = f"cuda:{local}"
for tag, cfg, n in (("C3 unet32+pixelshuffle batch=32/GPU 1080p bf16", dict(name="unet", base_channels=32, num_classes=3, enhance=True, upsampler="pixelshuffle"), 32),
                    ("C3' unet32 bilinear batch=32/GPU 1080p bf16", dict(name="unet", base_channels=32, num_classes=3, enhance=True), 32),
                    ("C4 retinexnet batch=32/GPU (256 over 8 GPUs) 1080p bf16", dict(name="retinexnet"), 32)):
    torch.cuda.reset_peak_memory_stats()
    m = mon_amd.MODELS.build(seed=1, compute_dtype="bf16", **cfg).to(dev)
    lo, hi = mon_amd.sharding.shard_range(n * world, rank, world)          # this rank's frames of the global batch
    x = torch.rand(hi - lo, 3, 1080, 1920, device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
    for _ in range(2):
        out = m.forward({"image": x})
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        out = m.forward({"image": x})
    torch.cuda.synchronize()
    dt = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"config": tag, "n_gpus": world, "frames_per_s": round(n * world / dt.item(), 1), "ms_per_batch": round(dt.item() * 1e3, 1),
                          "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "finite": bool(torch.isfinite(out["enhanced"]).all())}), flush=True)
    del m, x, out
    torch.cuda.empty_cache()
if world > 1:
    dist.destroy_process_group()
