"""The sharded-map matching exchange of vi_slam_amd.dist over RCCL (backend "nccl"): per-shard top-2 through the HIP
entry points, all-gather, (dist, idx) merge -- must equal one scan over the whole map and the CPU oracle.
Two ranks with one GPU each need two GPUs (skipped on a one-GPU box; the same function over gloo runs in
test_dist_cpu.py); the one-rank group runs everywhere and executes the same RCCL collective and merge."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, nq, nt, ret, force=False):
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
    calls = []
    if force:  # count the collectives the call issues: the forced path must not return before the exchange
        real = dist.all_gather_into_tensor

        def counted(out, inp, group=None):
            calls.append((tuple(out.shape), out.device.type))
            return real(out, inp, group=group)
        dist.all_gather_into_tensor = counted
    idx, d = vdist.sharded_match_knn2(qd, torch.from_numpy(t[b:e].copy()).to(dev), b, force_collective=force)
    if force:
        assert calls == [((world, 2, nq, 2), "cuda")], calls
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


def test_one_rank_rccl_group_runs_all_gather_and_merge():
    """A one-rank RCCL group on the one GPU every box has: group initialisation is the child's first GPU work, the forced
    path issues all_gather_into_tensor on device tensors and vfe_dev_merge_knn2, and the result equals the single scan and
    the oracle (a non-zero base index would shift every reported row, so the shard starts at row 0 of the map it is)."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_worker, args=(0, 1, port, 500, 200_000, ret, True))
    p.start()
    p.join(240)
    assert p.exitcode == 0
    assert ret.get(0) is True, dict(ret)
