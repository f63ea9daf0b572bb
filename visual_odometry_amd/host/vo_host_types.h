// vo_host_types.h — OpenCV-free carriers for the host-side mirror of the reference's interfaces.
//
// The reference passes cv::Mat / cv::KeyPoint / cv::DMatch / cv::detail::ImageFeatures /
// cv::detail::MatchesInfo across its features / matcher / sfm seam
// (reference: src/matcher/matcher.h:11-13, src/sfm/triangulation.h:9-20,
// src/visual_odometry/vo_frame.h:14-45).  OpenCV does not exist in this image, so the mirror uses
// the plain types below with the same member names and memory layouts (KeyPoint = 28 bytes,
// DMatch = 16 bytes, identical to cv's and to include/vo_types.h).  INTEGRATION.md shows the
// cv-typed adapter a maintainer compiles where OpenCV is installed; it forwards to the same C ABI.
#ifndef VO_HOST_TYPES_H
#define VO_HOST_TYPES_H

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vo_hip.h"

namespace vo {

struct Point2f {
    float x = 0, y = 0;
};
struct Point3d {
    double x = 0, y = 0, z = 0;
};
struct Size {
    int width = 0, height = 0;
};
struct KeyPoint {  // layout of cv::KeyPoint
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
struct DMatch {  // layout of cv::DMatch
    int queryIdx = -1, trainIdx = -1, imgIdx = -1;
    float distance = 3.402823466e+38F;
};
static_assert(sizeof(KeyPoint) == sizeof(vo_keypoint) && sizeof(DMatch) == sizeof(vo_dmatch), "ABI layouts");

// dense row-major double matrix (where the reference holds a CV_64F cv::Mat: K, E, R, t, poses)
struct Matd {
    int rows = 0, cols = 0;
    std::vector<double> d;
    Matd() {}
    Matd(int r, int c, double v = 0) : rows(r), cols(c), d((size_t)r * c, v) {}
    static Matd eye(int r, int c) {
        Matd m(r, c);
        for (int i = 0; i < r && i < c; i++) m(i, i) = 1;
        return m;
    }
    double& operator()(int r, int c) { return d[(size_t)r * cols + c]; }
    double operator()(int r, int c) const { return d[(size_t)r * cols + c]; }
    bool empty() const { return d.empty(); }
    Matd t() const {
        Matd m(cols, rows);
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++) m(c, r) = (*this)(r, c);
        return m;
    }
};
Matd operator*(const Matd& a, const Matd& b);
Matd operator+(const Matd& a, const Matd& b);
Matd operator*(double s, const Matd& a);
Matd operator-(const Matd& a);
Matd hconcat(const Matd& a, const Matd& b);

// 8-bit single-channel image (the reference hands 8UC3 imread output whose three channels are
// equal for KITTI's gray PNGs; cv converts it back to the same gray values: SURVEY.md A.1)
struct Image {
    int width = 0, height = 0;
    std::vector<uint8_t> px;
    // identity of the pixel content: copies share it, every newly made image gets a fresh one.  Lets
    // the tracker see that a frame is the one it tracked into last time (its pyramid is still on the
    // device).  Do not change px after the image was handed to a VOFrame.
    uint64_t id = next_id();
    bool empty() const { return px.empty(); }
    static uint64_t next_id();
};

// N x 32 CV_8U descriptor matrix
struct Descriptors {
    int rows = 0, cols = 0;
    std::vector<uint8_t> d;
};

// N x 1 CV_8U mask, read by the reference through at<bool>(i)
struct Mask {
    std::vector<uint8_t> d;
    bool at(int i) const { return d[(size_t)i] != 0; }
    int rows() const { return (int)d.size(); }
};

struct ImageFeatures {  // cv::detail::ImageFeatures
    int img_idx = 0;
    Size img_size;
    std::vector<KeyPoint> keypoints;
    Descriptors descriptors;
};

struct MatchesInfo {  // cv::detail::MatchesInfo
    int src_img_idx = -1, dst_img_idx = -1;
    std::vector<DMatch> matches;
    std::vector<uint8_t> inliers_mask;
    int num_inliers = 0;
    Matd H;
    double confidence = 0;
};

// what cv::Exception is to the reference: an error raised by the device library
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// One lazily created vo_ctx per host thread for the free functions (matcher, updatePose,
// triangulate); the device index comes from VO_HIP_DEVICE (default 0).  Throws vo::Error when no
// HIP device is usable: there is no CPU fallback.
vo_ctx* default_ctx();
void check(vo_ctx* ctx, int rc);

}  // namespace vo

#endif
