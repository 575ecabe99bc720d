// Needs a GPU.  vi_slam::BatchFrontEnd against separate engines: six streams (three distinct sequences, each twice) in two
// shards -- both on device 0, which exercises the sharding on a one-GPU machine -- must give every stream, frame by frame,
// exactly the points a SLAMEngine with the VINS-Mono adapter produces for that stream alone (the harness shape of the
// reference's tests/e2e/e2e_test_fixture.cpp:115-171, one engine per stream).  Both submission paths are used: frames in
// caller memory (with a row step larger than the width) and frames written into the pinned staging memory.
#include "gpu_test_frames.hpp"
#include "slam/batch_frontend.hpp"
#include "slam/slam_engine.hpp"
#include "slam/adapters/vins_mono_adapter.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>

using namespace vi_slam;

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

int main() {
    const int W = 752, H = 480, S = 6, D = 3, T = 7, N = 150, STEP = 768;
    { std::ofstream f("/tmp/vsa_batch_cfg.yaml"); f << "max_cnt: 150\nmin_dist: 30\nF_threshold: 0\nequalize: 0\npyramid_levels: 3\n"; }
    { std::ofstream f("/tmp/vsa_batch_calib.yaml"); f << "cam0: {}\n"; }
    int b, e;
    BatchFrontEnd::shardRange(64, 3, 8, b, e);
    CHECK(b == 24 && e == 32);
    BatchFrontEnd::shardRange(7, 0, 3, b, e);
    CHECK(b == 0 && e == 3);
    BatchFrontEnd::shardRange(7, 2, 3, b, e);
    CHECK(b == 5 && e == 7);

    BatchFrontEndConfig cfg;
    cfg.width = W; cfg.height = H; cfg.streams = S; cfg.maxFeatures = N; cfg.pyramidLevels = 3; cfg.minDistance = 30.0;
    cfg.devices = {0, 0};
    cfg.copyThreads = 2;
    BatchFrontEnd batch;
    CHECK(batch.open(cfg));
    CHECK(batch.shards() == (int)cfg.devices.size());
    {
        BatchFrontEndConfig bad = cfg;
        bad.devices = {0, 63};
        BatchFrontEnd other;
        CHECK(!other.open(bad) && other.lastError().find("out of range") != std::string::npos);
    }

    std::unique_ptr<SLAMEngine> engines[S];
    for (int s = 0; s < S; s++) {
        engines[s].reset(new SLAMEngine);
        CHECK(engines[s]->selectFramework(SLAMFrameworkType::VINS_MONO));
        CHECK(engines[s]->initialize("/tmp/vsa_batch_cfg.yaml"));
        CHECK(engines[s]->loadCalibration("/tmp/vsa_batch_calib.yaml"));
    }
    std::vector<BatchStreamResult> res;
    int prevTicket = -1;
    std::vector<std::vector<std::vector<float>>> want(T, std::vector<std::vector<float>>(S));
    std::vector<std::vector<uint8_t>> padded(S, std::vector<uint8_t>((size_t)STEP * H, 7));
    long totalPts = 0;
    for (int t = 0; t < T; t++) {
        /* frames of this time step: stream s plays sequence s % D */
        std::vector<std::vector<uint8_t>> img(D);
        for (int d = 0; d < D; d++) img[d] = testFrame(W, H, t, 1000u + 17u * d);
        for (int s = 0; s < S; s++) {
            engines[s]->processImage(ImageView(img[s % D].data(), H, W, W, 1), 1000000000LL + t * 33333333LL);
            CHECK(engines[s]->getFrontEndError().empty());
            auto* a = dynamic_cast<VINSMonoAdapter*>(engines[s]->framework());
            CHECK(a);
            want[t][s] = a->getTrackedPoints();
        }
        int ticket;
        if (t % 2 == 0) { /* caller memory, rows STEP bytes apart */
            std::vector<ImageView> views;
            for (int s = 0; s < S; s++) {
                for (int y = 0; y < H; y++) std::memcpy(&padded[s][(size_t)y * STEP], &img[s % D][(size_t)y * W], W);
                views.emplace_back(padded[s].data(), H, W, STEP, 1);
            }
            ticket = batch.submit(views.data(), S, t);
        } else {          /* frames produced in place in the pinned staging memory */
            const size_t pitch = batch.stagingPitch();
            for (int s = 0; s < S; s++) {
                uint8_t* dst = batch.staging(s);
                CHECK(dst);
                for (int y = 0; y < H; y++) std::memcpy(dst + (size_t)y * pitch, &img[s % D][(size_t)y * W], W);
            }
            ticket = batch.submitStaged(t);
        }
        CHECK(ticket == t);
        /* results are collected one submission late: the step of set t runs while set t + 1 is prepared */
        if (prevTicket >= 0) {
            CHECK(batch.wait(prevTicket, res));
            for (int s = 0; s < S; s++) {
                const std::vector<float>& w = want[prevTicket][s];
                CHECK(res[s].total * 2 == (int)w.size());
                CHECK(std::memcmp(res[s].xy, w.data(), w.size() * sizeof(float)) == 0);
                CHECK(res[s].tracked + res[s].detected == res[s].total);
                if (prevTicket > 0) CHECK(res[s].tracked > 40);
                totalPts += res[s].total;
            }
            CHECK(res[0].total == res[3].total); /* copies of one sequence */
        }
        prevTicket = ticket;
    }
    CHECK(batch.wait(prevTicket, res));
    for (int s = 0; s < S; s++) {
        const std::vector<float>& w = want[prevTicket][s];
        CHECK(res[s].total * 2 == (int)w.size() && std::memcmp(res[s].xy, w.data(), w.size() * sizeof(float)) == 0);
    }
    CHECK(!batch.wait(prevTicket - 3, res)); /* too old: its slot has been reused */
    CHECK(!batch.wait(prevTicket + 1, res)); /* never submitted */
    /* reset: the next set is a first frame again */
    CHECK(batch.reset());
    std::vector<ImageView> views;
    std::vector<uint8_t> first = testFrame(W, H, 0, 1000u);
    for (int s = 0; s < S; s++) views.emplace_back(first.data(), H, W, W, 1);
    const int t2 = batch.submit(views.data(), S, 0);
    CHECK(t2 == prevTicket + 1 && batch.wait(t2, res));
    CHECK(res[0].tracked == 0 && res[0].total * 2 == (int)want[0][0].size());
    CHECK(std::memcmp(res[0].xy, want[0][0].data(), want[0][0].size() * sizeof(float)) == 0);
    /* wrong frame size is refused */
    views[2] = ImageView(first.data(), H - 1, W, W, 1);
    CHECK(batch.submit(views.data(), S, 0) < 0);
    batch.close();
    std::printf("batch front-end: %d streams x %d frames in 2 shards equal %d separate engines (%ld points compared): all checks passed\n",
                S, T, S, totalPts);
    return 0;
}
