"""The C++ sharded matcher (vi_slam::ShardedMatcher: per-device scan, RCCL all-gathers issued from C++, merge) on a GPU.
A one-GPU box runs it with one communicator rank -- the collective and the merge execute all the same; with two or more
GPUs the map is really split."""
import numpy as np
import pytest

from oracle import oracle as ora
from vi_slam_amd import engine as eng
from vi_slam_amd import synth

pytestmark = pytest.mark.gpu


def _sets(nq, nt):
    q = synth.descriptors(41, nq)
    t = synth.descriptors(synth.MAP_SEED + 9, nt)
    for i in range(min(nq, 80)):
        t[(i * 1031 + 7) % nt] = synth.flip_bits(900 + i, q[i], i % 21)
    t[5] = t[nt - 3]  # equal rows far apart (in different shards whenever there are two): the lower row must win
    q[1] = t[5]
    return q, t


def _check(devices):
    nq, nt = 700, 260_001  # an odd row count: shards of unequal size
    q, t = _sets(nq, nt)
    m = eng.ShardedMatcher(devices)
    assert m.world == max(1, len(devices))
    assert "ncclAllGather" in m.collective and "RCCL" in m.collective
    m.set_map(t)
    idx, dist, good = m.match(q, 0.7)
    ridx, rdist = ora.match_knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert idx[1, 0] == 5 and dist[1, 0] == 0 and idx[1, 1] == nt - 3 and dist[1, 1] == 0
    assert np.array_equal(good, ora.ratio_test(ridx, rdist, 0.7).astype(bool))
    # a second, smaller query set through the same buffers, then a new map
    idx2, dist2, _ = m.match(q[:33], 0.7)
    assert np.array_equal(idx2, ridx[:33]) and np.array_equal(dist2, rdist[:33])
    m.set_map(t[:1000])
    idx3, dist3, _ = m.match(q[:50], 0.7)
    r3i, r3d = ora.match_knn2(q[:50], t[:1000])
    assert np.array_equal(idx3, r3i) and np.array_equal(dist3, r3d)
    assert m.last_ms > 0
    m.close()


def test_sharded_matcher_one_rank(cuda):
    _check((0,))


def test_sharded_matcher_two_ranks(cuda):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _check((0, 1))


def test_sharded_matcher_refuses_bad_device_lists(cuda):
    with pytest.raises(RuntimeError, match="listed twice"):
        eng.ShardedMatcher((0, 0))
    with pytest.raises(RuntimeError, match="out of range"):
        eng.ShardedMatcher((0, 63))
