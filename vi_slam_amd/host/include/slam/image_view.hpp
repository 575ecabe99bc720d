#ifndef VI_SLAM_AMD_IMAGE_VIEW_HPP
#define VI_SLAM_AMD_IMAGE_VIEW_HPP
/* Borrowed view of an 8-bit image: exactly the fields the reference's adapters read from the
 * cv::Mat they receive (data, rows, cols, step, channels; include/slam/i_slam_framework.hpp:63).
 * When OpenCV headers are available a cv::Mat converts implicitly, so reference call sites
 * compile unchanged. */
#include <cstddef>
#include <cstdint>
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define VI_SLAM_AMD_HAVE_OPENCV 1
#endif

namespace vi_slam {

struct ImageView {
    const uint8_t* data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0;   /* bytes per row */
    int nchannels = 1;
    ImageView() = default;
    ImageView(const uint8_t* d, int r, int c, size_t s, int ch) : data(d), rows(r), cols(c), step(s), nchannels(ch) {}
#ifdef VI_SLAM_AMD_HAVE_OPENCV
    ImageView(const cv::Mat& m)  // NOLINT: implicit on purpose
        : data(m.data), rows(m.rows), cols(m.cols), step(m.step), nchannels(m.channels()) {}
#endif
    bool empty() const { return data == nullptr || rows <= 0 || cols <= 0; }
    int channels() const { return nchannels; }
};

}  // namespace vi_slam
#endif
