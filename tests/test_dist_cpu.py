"""CPU, 2 processes over gloo: the sharded-map matching exchange (all-gather of per-shard top-2,
merge by (dist, idx)) and the stream sharding used by bench.py.  The per-shard scan and the merge
are the oracle's here (no GPU); on GPUs the same function runs the HIP entry points over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as ora
from vi_slam_amd import dist as vdist
from vi_slam_amd import synth


def test_shard_ranges_partition_exactly():
    for n in [0, 1, 7, 64, 10_000_000, 2001]:
        for world in [1, 2, 3, 4, 8]:
            r = [vdist.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1
    assert vdist.streams_for_rank(64, 3, 8) == list(range(24, 32))


def _oracle_match(q, t, base):
    i, d = ora.match_knn2(q.numpy(), t.numpy(), base)
    return torch.from_numpy(i), torch.from_numpy(d)


def _numpy_merge(all_idx, all_dist):
    s, nq, _ = all_idx.shape
    out_i = np.full((nq, 2), -1, np.int32)
    out_d = np.full((nq, 2), 0x7fffffff, np.int32)
    ai, ad = all_idx.numpy(), all_dist.numpy()
    for q in range(nq):
        cand = sorted((int(ad[r, q, k]), int(ai[r, q, k])) for r in range(s) for k in range(2) if ai[r, q, k] >= 0)[:2]
        for k, (d, i) in enumerate(cand):
            out_i[q, k], out_d[q, k] = i, d
    return torch.from_numpy(out_i), torch.from_numpy(out_d)


def _worker(rank, world, port, nq, nt, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q = synth.descriptors(21, nq)
    t = synth.descriptors(synth.MAP_SEED, nt)
    for i in range(40):
        t[(i * 131 + 7) % nt] = synth.flip_bits(i, q[i], i % 21)
    t[5] = t[nt - 3]  # equal descriptors in different shards
    b, e = vdist.shard_range(nt, rank, world)
    idx, d = vdist.sharded_match_knn2(torch.from_numpy(q), torch.from_numpy(t[b:e].copy()), b,
                                      match_fn=_oracle_match, merge_fn=_numpy_merge)
    ridx, rdist = ora.match_knn2(q, t)
    ok = np.array_equal(idx.numpy(), ridx) and np.array_equal(d.numpy(), rdist)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_match_over_gloo_equals_single_scan(world):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 120, 5000, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _forced_worker(port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    q = synth.descriptors(3, 50)
    t = synth.descriptors(4, 900)
    merges = []

    def merge(ai, ad):
        merges.append(tuple(ai.shape))
        return _numpy_merge(ai, ad)
    try:
        vdist.sharded_match_knn2(torch.from_numpy(q), torch.from_numpy(t), 11, match_fn=_oracle_match, merge_fn=merge,
                                 force_collective=True)
        ret["raised"] = False
    except RuntimeError:
        ret["raised"] = True   # no process group: the forced exchange cannot run and says so
    dist.init_process_group("gloo", rank=0, world_size=1)
    plain = vdist.sharded_match_knn2(torch.from_numpy(q), torch.from_numpy(t), 11, match_fn=_oracle_match, merge_fn=merge)
    assert merges == []         # one rank, not forced: the scan's result is returned as it is
    idx, d = vdist.sharded_match_knn2(torch.from_numpy(q), torch.from_numpy(t), 11, match_fn=_oracle_match, merge_fn=merge,
                                      force_collective=True)
    ret["merges"] = list(merges)
    ret["equal"] = bool(torch.equal(idx, plain[0]) and torch.equal(d, plain[1]))
    dist.destroy_process_group()


def test_forced_collective_in_a_one_rank_group():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_forced_worker, args=(port, ret))
    p.start()
    p.join(120)
    assert p.exitcode == 0
    assert ret["raised"] is True and ret["equal"] is True
    assert ret["merges"] == [(1, 50, 2)]
