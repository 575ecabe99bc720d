"""The allocation-free matching entry point (vfe_dev_match_knn2_noalloc) against vfe_dev_match_knn2 through the C ABI:
bit-identical top-2 lists on two calls in a row, the second with a smaller input, so that whatever the first call left
in the library's built-in scratch is shown not to leak into the second; oversized requests are rejected."""
import numpy as np
import pytest

from vi_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def _sets(nq, nt, seed):
    q = synth.descriptors(seed, nq)
    t = synth.descriptors(seed + 1, nt)
    for i in range(min(nq, nt, 64)):          # near-duplicates and an exact tie
        t[(i * 131 + 7) % nt] = synth.flip_bits(200 + i, q[i], i % 20)
    if nt > 200:
        t[11] = t[173]
    return q, t


def test_noalloc_matches_workspace_entry_point_twice(cuda):
    import torch
    before = torch.cuda.memory_allocated(cuda)
    # large first, then smaller; the third is a train set past the size from which a sample is searched first, so the
    # sample's lists and the per-query floors in the scratch are in use, and the fourth is small again after it
    for k, (nq, nt, base) in enumerate([(2000, 120000, 5000), (300, 4000, 17), (2000, 640_000, 900), (257, 700, 0)]):
        q, t = _sets(nq, nt, 300 + 2 * k)
        qd, td = torch.from_numpy(q).to(cuda), torch.from_numpy(t).to(cuda)
        ref_idx, ref_dist = capi.match_knn2(qd, td, base_index=base)
        out = (torch.full((nq, 2), -7, dtype=torch.int32, device=cuda), torch.full((nq, 2), -7, dtype=torch.int32, device=cuda))
        if k == 0:  # the first call on a device looks the scratch's address up; the lookup is not what is measured below
            capi.match_knn2_noalloc(qd, td, base_index=base, out=out)
        torch.cuda.synchronize()
        held = torch.cuda.memory_allocated(cuda)
        free_before = torch.cuda.mem_get_info(cuda)[0]
        idx, dist = capi.match_knn2_noalloc(qd, td, base_index=base, out=out)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated(cuda) == held  # nothing came from torch's allocator ...
        assert torch.cuda.mem_get_info(cuda)[0] == free_before  # ... and nothing from the device's (a hipMalloc inside the library)
        assert torch.equal(idx, ref_idx) and torch.equal(dist, ref_dist), f"call {k}"
        # the workspace path is itself checked against the CPU oracle in test_gpu_orb_match.py
        del qd, td, out, idx, dist, ref_idx, ref_dist
    assert torch.cuda.memory_allocated(cuda) <= before + (1 << 20)


def test_noalloc_empty_train_and_rejection(cuda):
    import ctypes as C
    import torch
    q = torch.from_numpy(synth.descriptors(5, 70)).to(cuda)
    idx, dist = capi.match_knn2_noalloc(q, torch.zeros((0, 32), dtype=torch.uint8, device=cuda))
    assert (idx.cpu().numpy() == -1).all() and (dist.cpu().numpy() == 0x7fffffff).all()
    # more queries than the scratch has list slots for: refused before anything is launched or read
    L = capi.lib()
    nq_big = 1_200_000
    assert L.vfe_match_workspace(nq_big, 64) > capi.MATCH_NOALLOC_BYTES
    rc = L.vfe_dev_match_knn2_noalloc(C.c_void_p(q.data_ptr()), nq_big, C.c_void_p(q.data_ptr()), 64, 0,
                                      C.c_void_p(idx.data_ptr()), C.c_void_p(dist.data_ptr()), None)
    assert rc == -4, rc  # VFE_E_CAPACITY
    assert b"scratch" in L.vfe_last_error()


def test_noalloc_at_the_capacity_edge(cuda):
    """The largest query set whose lists still fit the scratch for a sampled train set (found with vfe_match_workspace, as
    the header tells callers to) is served and agrees with the workspace entry point; the first size that does not fit
    is refused."""
    import ctypes as C
    import torch
    L = capi.lib()
    nt = 520_000
    assert L.vfe_match_workspace(2000, 16_000_000) <= capi.MATCH_NOALLOC_BYTES  # the relocalisation shape fits with room
    lo, hi = 2000, 200_000
    assert L.vfe_match_workspace(lo, nt) <= capi.MATCH_NOALLOC_BYTES < L.vfe_match_workspace(hi, nt)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if L.vfe_match_workspace(mid, nt) <= capi.MATCH_NOALLOC_BYTES:
            lo = mid
        else:
            hi = mid
    q, t = _sets(lo, nt, 900)
    qd, td = torch.from_numpy(q).to(cuda), torch.from_numpy(t).to(cuda)
    ref_idx, ref_dist = capi.match_knn2(qd, td, base_index=3)
    idx, dist = capi.match_knn2_noalloc(qd, td, base_index=3)
    assert torch.equal(idx, ref_idx) and torch.equal(dist, ref_dist)
    qh = torch.from_numpy(synth.descriptors(901, hi)).to(cuda)
    o = torch.empty((hi, 2), dtype=torch.int32, device=cuda)
    rc = L.vfe_dev_match_knn2_noalloc(C.c_void_p(qh.data_ptr()), hi, C.c_void_p(td.data_ptr()), nt, 0,
                                      C.c_void_p(o.data_ptr()), C.c_void_p(o.data_ptr()), None)
    assert rc == -4, rc
