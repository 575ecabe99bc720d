"""The relocalisation workload with known answers: a 256-bit map of seeded random rows in which every query has a planted
near-duplicate at a known global row.  Used by bench.py's `reloc` leg and by the full-size GPU test; host-side only."""
import numpy as np

from . import synth

MAP_ROWS, N_QUERIES = 10_000_000, 2000


def planted_position(i, map_rows=MAP_ROWS):
    return (i * 4999 + 1234567) % map_rows  # 4999 is coprime to 10^7: distinct rows


def map_shard(q, begin, end, map_rows=MAP_ROWS):
    """Rows [begin, end) of the map: the seeded stream, with query i's near-duplicate (i % 21 flipped bits) at
    planted_position(i) where that row falls into the shard."""
    shard = synth.descriptors(synth.MAP_SEED, end - begin, first=begin)
    for i in range(len(q)):
        p = planted_position(i, map_rows)
        if begin <= p < end:
            shard[p - begin] = synth.flip_bits(i, q[i], i % 21)
    return shard


def verify_planted(q, idx, dist_, map_rows=MAP_ROWS):
    """Known answers of the relocalisation workload.  Query i has a planted row (i % 21 flipped bits) at
    planted_position(i); ORB queries can be near-duplicates of each other, so the expected neighbours of query i are the
    best two of ALL planted rows by (distance, global row), computed here on the host from the nq x nq distances.
    Unplanted rows are uniform random bits (about 128 +- 8 bits away from anything; the closest of 10 M is ~85 bits away),
    so they cannot enter a top-2 whose distances are below 60.
    -> (all as expected, number of queries whose top-1 is their own planted row)."""
    n = len(q)
    planted = np.stack([synth.flip_bits(i, q[i], i % 21) for i in range(n)])
    pos = np.array([planted_position(i, map_rows) for i in range(n)], np.int64)
    q64, p64 = q.view(np.uint64), planted.view(np.uint64)
    ok, own = True, 0
    for i0 in range(0, n, 250):
        d = np.bitwise_count(q64[i0:i0 + 250, None, :] ^ p64[None, :, :]).sum(axis=2).astype(np.int64)  # [250, n]
        key = d * (1 << 32) + pos[None, :]
        best = np.argsort(key, axis=1)[:, :2]
        for r in range(d.shape[0]):
            i = i0 + r
            j0, j1 = best[r]
            ok &= bool(idx[i, 0] == pos[j0] and dist_[i, 0] == d[r, j0])
            if d[r, j1] < 60:
                ok &= bool(idx[i, 1] == pos[j1] and dist_[i, 1] == d[r, j1])
            own += int(idx[i, 0] == pos[i])
    return ok, own
