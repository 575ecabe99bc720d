"""Two ranks over RCCL (backend "nccl"), one GPU each: the sharded-map matching exchange of vi_slam_amd.dist --
per-shard top-2 through the HIP entry points, all-gather, (dist, idx) merge -- must equal one scan over the whole map
and the CPU oracle.  Needs two GPUs; skipped on a one-GPU box (the same function over gloo runs in test_dist_cpu.py)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, nq, nt, ret):
    import torch
    import torch.distributed as dist
    from vi_slam_amd import capi, dist as vdist, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    q = synth.descriptors(21, nq)
    t = synth.descriptors(synth.MAP_SEED, nt)
    for i in range(60):
        t[(i * 1031 + 7) % nt] = synth.flip_bits(i, q[i], i % 21)
    t[5] = t[nt - 3]  # equal descriptors in different shards: the lower global index must win
    b, e = vdist.shard_range(nt, rank, world)
    qd = torch.from_numpy(q).to(dev)
    idx, d = vdist.sharded_match_knn2(qd, torch.from_numpy(t[b:e].copy()).to(dev), b)
    full_idx, full_d = capi.match_knn2(qd, torch.from_numpy(t).to(dev))
    ok = torch.equal(idx, full_idx) and torch.equal(d, full_d)
    if rank == 0:
        from oracle import oracle as ora
        ridx, rdist = ora.match_knn2(q, t)
        ok = ok and np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rdist)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_match_over_rccl_equals_single_scan():
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:  # counting devices does not initialise the GPU in this process
        pytest.skip("needs two GPUs")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 500, 200_000, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(2)), dict(ret)
