/* CPU oracle (test infrastructure only): brute-force Hamming kNN (k=2) and Lowe ratio test.
 * Restates cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, 2) (OpenCV 4.5 batchDistance,
 * K>0 branch) -- the matcher the reference's OpenVINS descriptor mode names
 * (docs/reference/15_openvins_smartphone_config.md:99-105, knn_ratio 0.70). */
#include "ora_common.h"
#include "vfe_oracle.h"

extern "C" void ora_match_knn2(const uint8_t* query, int nq, const uint8_t* train, int nt,
                               int base_index, int* out_idx, int* out_dist) {
    for (int q = 0; q < nq; q++) {
        uint64_t a[4];
        std::memcpy(a, query + (size_t)q * 32, 32);
        int d0 = INT32_MAX, d1 = INT32_MAX, i0 = -1, i1 = -1;
        for (int t = 0; t < nt; t++) {
            uint64_t b[4];
            std::memcpy(b, train + (size_t)t * 32, 32);
            int d = __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
                    __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
            /* strict <: ties keep the lower train index; result sorted by (dist, idx) */
            if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = base_index + t; }
            else if (d < d1) { d1 = d; i1 = base_index + t; }
        }
        out_idx[2 * q] = i0; out_idx[2 * q + 1] = i1;
        out_dist[2 * q] = d0; out_dist[2 * q + 1] = d1;
    }
}

extern "C" void ora_ratio_test(const int* idx, const int* dist, int nq, float ratio, uint8_t* good) {
    for (int q = 0; q < nq; q++)
        good[q] = (idx[2 * q] >= 0 && idx[2 * q + 1] >= 0 && (float)dist[2 * q] < ratio * (float)dist[2 * q + 1]) ? 1 : 0;
}
