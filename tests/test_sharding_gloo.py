"""CPU, world_size 2 over gloo: the multi-GPU inference path shards frames with no data-path
collective; gather brings uneven shards back in order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mon_amd.sharding import gather_frames, scatter_frames, shard_range
    frames = torch.arange(n_total * 6, dtype=torch.float32).reshape(n_total, 1, 2, 3)
    local = scatter_frames(frames, rank, world)
    lo, hi = shard_range(n_total, rank, world)
    assert local.shape[0] == hi - lo
    processed = local * 2 + 1          # stands in for the per-rank enhancement
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    out = gather_frames(processed, n_total, dst=0)
    if rank == 0:
        ret["ok"] = bool(torch.equal(out, frames * 2 + 1))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_gather_world2():
    for n_total in (5, 8):
        mgr = mp.Manager()
        ret = mgr.dict()
        port = _free_port()
        mp.spawn(_worker, args=(2, port, n_total, ret), nprocs=2, join=True)
        assert ret.get("ok") is True


def _grad_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mon_amd.sharding import allreduce_gradients
    g = torch.arange(10, dtype=torch.float32) * (rank + 1)       # rank r holds (r+1) * [0..9]
    scale = allreduce_gradients(g)
    mean = g * scale
    expect = torch.arange(10, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    ret[rank] = bool(torch.allclose(mean, expect)) and scale == 1.0 / world
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_mean_world2():
    """Training path: the flat-buffer sum all-reduce + 1/world factor equals DDP's gradient mean."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_grad_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret.get(0) is True and ret.get(1) is True


def test_gradient_allreduce_without_group_is_identity():
    from mon_amd.sharding import allreduce_gradients
    g = torch.ones(4)
    assert allreduce_gradients(g) == 1.0 and torch.equal(g, torch.ones(4))


def test_bench_gpus_n_starts_n_ranks_command():
    """`python bench.py --gpus N` without a launcher re-launches itself as N ranks (torch.distributed.run on 127.0.0.1) before
    any GPU call; the JSON line's n_gpus is the children's WORLD_SIZE (checked on the GPU box by test_bench_two_ranks_one_device)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    cmd = b._spawn_cmd(4, ["--gpus", "4", "--steps", "2"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert cmd[cmd.index("--master-port") + 2].endswith("bench.py")
    src = open(os.path.join(root, "bench.py")).read()
    assert src.index("_spawn_ranks(args.gpus)") < src.index("import torch\n")      # spawn happens before torch / HIP is touched
    # the board probe's text parser and the CPU-baseline thread count (affinity mask cut to the cgroup quota)
    t = "GPU[0]\t\t: sclk clock level: 1: (1828Mhz)\nGPU[0]\t\t: Max Graphics Package Power (W): 1400.0\nGPU[0]\t\t: Current Socket Graphics Package Power (W): 1399.0\n"
    assert b._parse_rocm_smi(t) == {"socket_power_W": 1399.0, "power_cap_W": 1400.0, "sclk_MHz": 1828} and b._parse_rocm_smi("no such tool") is None
    assert 1 <= b._usable_cpus() <= (os.cpu_count() or 1)
    calls = []
    assert b._board_state(lambda: calls.append(1), lambda: calls.append("drained"), 1e9) is None or calls       # no rocm-smi here (or a real one): never raises
