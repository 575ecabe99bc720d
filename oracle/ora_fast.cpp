/* CPU oracle (test infrastructure only): FAST-9/16 with non-max suppression, and the
 * grid-bucketed FAST detector the reference's OpenVINS configuration names
 * (config/openvins.yaml:29-37: fast_threshold, grid_x, grid_y, min_px_dist, num_pts;
 * include/slam/adapters/openvins_adapter.hpp:29-32).  FAST itself restates
 * cv::FAST(TYPE_9_16) from OpenCV 4.5, which cv::ORB (orbslam3_adapter.cpp:76-79) runs
 * on every pyramid level. */
#include "ora_common.h"
#include "vfe_oracle.h"

using namespace ora;

namespace {
const int OFF16[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* Corner test: >= 9 contiguous circle pixels all brighter than v+t or all darker than v-t. */
bool is_corner(const int* d /*v - circle, 16*/, int t) {
    /* darker-circle run: d > t ; brighter-circle run: d < -t */
    int run_d = 0, run_b = 0;
    for (int k = 0; k < 16 + 8; k++) {
        int v = d[k & 15];
        run_d = (v > t) ? run_d + 1 : 0;
        run_b = (v < -t) ? run_b + 1 : 0;
        if (run_d >= 9 || run_b >= 9) return true;
    }
    return false;
}

/* cornerScore<16>: the largest threshold for which the pixel is still a corner. */
int corner_score(const int* d16, int threshold) {
    int d[25];
    for (int k = 0; k < 25; k++) d[k] = d16[k & 15];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]); a = std::min(a, d[k + 5]); a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]); a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]); b = std::max(b, d[k + 4]); b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]); b = std::max(b, d[k + 7]); b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

/* score map of a w x h image: 0 for non-corners, cornerScore for corners (x,y in [3, dim-4]) */
void score_map(const uint8_t* img, int w, int h, int step, int threshold, std::vector<int>& sc) {
    sc.assign((size_t)w * h, 0);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int v = img[(size_t)y * step + x], d[16];
            for (int k = 0; k < 16; k++) d[k] = v - img[(size_t)(y + OFF16[k][1]) * step + x + OFF16[k][0]];
            if (is_corner(d, threshold)) sc[(size_t)y * w + x] = corner_score(d, threshold);
        }
}
}  // namespace

extern "C" int ora_fast9(const uint8_t* img, int w, int h, int step, int threshold, int nms,
                         int* kps, int cap) {
    std::vector<int> sc;
    score_map(img, w, h, step, threshold, sc);
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            /* a corner's score is >= threshold, but can be 0 only if threshold == 0;
             * use the corner test result via a second map-free rule: score>0 or thr==0&&corner */
            int s = sc[(size_t)y * w + x];
            bool corner = s > 0;
            if (!corner && threshold == 0) {
                int v = img[(size_t)y * step + x], d[16];
                for (int k = 0; k < 16; k++) d[k] = v - img[(size_t)(y + OFF16[k][1]) * step + x + OFF16[k][0]];
                corner = is_corner(d, 0);
            }
            if (!corner) continue;
            if (nms) {
                bool ok = true;
                for (int dy = -1; dy <= 1 && ok; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if (!dx && !dy) continue;
                        if (!(s > sc[(size_t)(y + dy) * w + x + dx])) { ok = false; break; }
                    }
                if (!ok) continue;
            }
            if (n >= cap) return -1;
            kps[3 * n] = x; kps[3 * n + 1] = y; kps[3 * n + 2] = s; ++n;
        }
    return n;
}

extern "C" int ora_fast_grid(const uint8_t* img, int w, int h, int step, int threshold,
                             int grid_x, int grid_y, int num_features, int min_px_dist,
                             const float* seeds, int n_seed,
                             float* pts_xy, int* scores, int cap) {
    const int needed = num_features - n_seed;
    if (needed <= 0 || grid_x <= 0 || grid_y <= 0) return 0;
    const int size_x = w / grid_x, size_y = h / grid_y;
    if (size_x < 7 || size_y < 7) return 0;
    const int per_cell = (int)((double)needed / (double)(grid_x * grid_y)) + 1;
    /* occupancy grid at min_px_dist resolution, pre-marked by the seeds (existing tracks) */
    const int ow = (int)((float)w / (float)min_px_dist), oh = (int)((float)h / (float)min_px_dist);
    std::vector<uint8_t> occ((size_t)std::max(ow, 0) * std::max(oh, 0), 0);
    for (int s = 0; s < n_seed; s++) {
        int gx = (int)(seeds[2 * s] / (float)min_px_dist), gy = (int)(seeds[2 * s + 1] / (float)min_px_dist);
        if (gx < 0 || gy < 0 || gx >= ow || gy >= oh) continue;
        occ[(size_t)gy * ow + gx] = 1;
    }
    int n = 0;
    std::vector<int> kps((size_t)size_x * size_y * 3);
    for (int r = 0; r < grid_x * grid_y; r++) {
        int x0 = (r % grid_x) * size_x, y0 = (r / grid_x) * size_y;
        int cnt = ora_fast9(img + (size_t)y0 * step + x0, size_x, size_y, step, threshold, 1, kps.data(), size_x * size_y);
        std::vector<int> order(cnt);
        for (int i = 0; i < cnt; i++) order[i] = i;
        /* response descending; ties keep raster order */
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return kps[3 * a + 2] > kps[3 * b + 2]; });
        for (int i = 0; i < cnt && i < per_cell; i++) {
            int k = order[i];
            float x = (float)(kps[3 * k] + x0), y = (float)(kps[3 * k + 1] + y0);
            int gx = (int)(x / (float)min_px_dist), gy = (int)(y / (float)min_px_dist);
            if (gx < 0 || gy < 0 || gx >= ow || gy >= oh) continue;
            if (occ[(size_t)gy * ow + gx]) continue;
            occ[(size_t)gy * ow + gx] = 1;
            if (n >= cap) return -1;
            pts_xy[2 * n] = x; pts_xy[2 * n + 1] = y; scores[n] = kps[3 * k + 2]; ++n;
        }
    }
    return n;
}
