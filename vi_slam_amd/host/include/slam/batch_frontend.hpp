#ifndef VI_SLAM_AMD_BATCH_FRONTEND_HPP
#define VI_SLAM_AMD_BATCH_FRONTEND_HPP
/* Many independent live streams through the GPU front-end at once, from C++.
 *
 * The reference's harness drives one engine with one stream, one frame per call (tests/e2e/e2e_test_fixture.cpp:115-171,
 * src/slam/slam_engine.cpp:94-119).  A machine that serves many cameras wants the same per-stream result -- the
 * VINS-style step of `vfe_process_frame`: pyramid, KLT from the previous frame, Shi-Tomasi refill -- for all streams of
 * a GPU in one launch sequence (`vfe_dev_frontend_step`, include/vfe.h), and the streams spread over the GPUs of the
 * node.  BatchFrontEnd does that:
 *   - streams are split over the configured devices in contiguous blocks (the same rule as the Python side's
 *     shard_range: sizes differ by at most one, lower shards get the larger blocks); no device-to-device traffic;
 *   - per device one host thread owns all HIP calls, a copy stream and a compute stream; frames go through pinned
 *     staging memory (three slots per stream), so the upload of frame set i + 1 runs beside the step on set i;
 *   - results (points, counts) come back through pinned memory and stay valid until three submissions later.
 * Every stream's points equal what a SLAMEngine with the VINS-Mono adapter produces for that stream alone (same
 * kernels, batch of one), which tests/cpp/test_batch_frontend.cpp checks.
 * There is no CPU fallback: open() fails without a usable GPU. */
#include "slam/image_view.hpp"
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace vi_slam {

struct BatchFrontEndConfig {
    int width = 1920, height = 1080;   /* every stream has this frame size (8-bit gray) */
    int streams = 64;
    int maxFeatures = 1000;            /* `max_cnt` */
    int pyramidLevels = 4;             /* LK levels including level 0 */
    int winSize = 21;
    double quality = 0.01;
    double minDistance = 20.0;         /* `min_dist` */
    std::vector<int> devices;          /* HIP device ordinals, one shard of streams each (an ordinal may repeat: two
                                        * shards on one GPU); empty = device 0 */
    int copyThreads = 4;               /* host threads per shard that repack caller frames into pinned memory */
};

struct BatchStreamResult {
    const float* xy = nullptr;  /* total (x, y) pairs: surviving tracks first, then new corners by descending score */
    int total = 0, tracked = 0, detected = 0;
};

class BatchFrontEnd {
public:
    BatchFrontEnd();
    ~BatchFrontEnd();
    BatchFrontEnd(const BatchFrontEnd&) = delete;
    BatchFrontEnd& operator=(const BatchFrontEnd&) = delete;

    bool open(const BatchFrontEndConfig& cfg);
    void close();
    bool isOpen() const;
    const BatchFrontEndConfig& config() const;
    int shards() const;
    /* streams [begin, end) of shard `rank` out of `world` */
    static void shardRange(int n, int rank, int world, int& begin, int& end);

    /* Pinned staging memory of `stream` for the NEXT submission: a caller that can produce frames in place (a decoder, a
     * receiver) writes rows of stagingPitch() bytes there and calls submitStaged(); no further host copy is made. */
    uint8_t* staging(int stream);
    size_t stagingPitch() const;
    /* Enqueue upload + front-end step for the frame set in the staging slot.  Returns the ticket (0, 1, 2, ...) or -1. */
    int submitStaged(int64_t timestampNs);
    /* The same for frames in caller memory (`count` must equal config().streams, sizes must match): the frames are
     * repacked into the staging slot by the shards' copy threads before the call returns -- the images are borrowed for
     * the duration of the call only, like the image of ISLAMFramework::processImage. */
    int submit(const ImageView* frames, int count, int64_t timestampNs);
    /* Block until the step of `ticket` is done and hand out every stream's points.  At most two tickets may be
     * outstanding beyond the one waited for; results stay valid until the third submission after theirs. */
    bool wait(int ticket, std::vector<BatchStreamResult>& out);
    /* forget all tracks: the next submission is a first frame again (detect only) */
    bool reset();
    std::string lastError() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace vi_slam
#endif
