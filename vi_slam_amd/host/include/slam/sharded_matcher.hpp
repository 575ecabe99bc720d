#ifndef VI_SLAM_AMD_SHARDED_MATCHER_HPP
#define VI_SLAM_AMD_SHARDED_MATCHER_HPP
/* Relocalisation-style matching against a map that is split over the GPUs of the node, from C++.
 *
 * The reference names the matcher only in its configuration (`knn_ratio`, docs/reference/15_openvins_smartphone_config.md:105)
 * and has no multi-device code at all; a map of millions of 256-bit descriptors wants one.  ShardedMatcher keeps rows
 * [shardRange(n, r, world)) of the map on device r (contiguous blocks, sizes differ by at most one, lower ranks larger:
 * the same rule as BatchFrontEnd and the Python side), replicates the query set, scans every shard for the per-query
 * top-2 (`vfe_dev_match_knn2`, global row indices), exchanges the per-shard lists with ONE grouped pair of RCCL
 * all-gathers over xGMI (2 x nq x 2 x 4 bytes per rank: 32 000 B at 2000 queries) and merges them by (distance, global
 * row) on every device (`vfe_dev_merge_knn2`) -- the order a single scan over the whole map produces, ties included.
 * One process drives all devices (ncclCommInitAll: one communicator rank per device); RCCL is loaded at run time, so the
 * host library has no link-time dependency on it.  No CPU fallback. */
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace vi_slam {

class ShardedMatcher {
public:
    ShardedMatcher();
    ~ShardedMatcher();
    ShardedMatcher(const ShardedMatcher&) = delete;
    ShardedMatcher& operator=(const ShardedMatcher&) = delete;

    /* devices: distinct HIP ordinals, one communicator rank each (empty = device 0) */
    bool open(const std::vector<int>& devices);
    void close();
    bool isOpen() const;
    int world() const;
    /* copy the map (n rows of 32 bytes, host memory) to the devices, one contiguous block of rows each */
    bool setMap(const uint8_t* descriptors, long long n);
    long long mapRows() const;
    /* kNN-2 of `nq` query descriptors (host memory) against the whole map.  idx / dist: nq x 2 (absent neighbours: -1 /
     * 0x7fffffff); good (may be null): Lowe's ratio test d0 < ratio * d1.  Blocks until the result is in the caller's
     * arrays. */
    bool match(const uint8_t* queries, int nq, int* idx, int* dist, float ratio, uint8_t* good);
    /* e.g. "ncclAllGather x2 grouped (RCCL 2.26.6, 1 rank)" once open */
    std::string collective() const;
    std::string lastError() const;
    /* milliseconds the last match() spent between the first enqueue and the last device's completion */
    double lastMatchMs() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace vi_slam
#endif
