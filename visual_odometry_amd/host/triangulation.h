// triangulation.h — two-view triangulation helpers of the host mirror.
//
// Same free-function names and argument order as the reference's src/sfm/triangulation.h:9-20, so
// updatePose() and a caller's own code read as they do there; the matrices are vo::Matd (row-major
// doubles) instead of cv::Mat, and the heavy call (triangulate) runs on the GPU through vo_triangulate.
#ifndef VO_HOST_TRIANGULATION_H
#define VO_HOST_TRIANGULATION_H

#include <cstdint>
#include <vector>

#include "vo_frame.h"

// K * [R^T | -R^T t] for a camera-to-world pose [R | t] (3 x 4); 3 x 4 result.  Host arithmetic
// (vo_projection_matrix), reference: triangulation.cpp:34-44.
vo::Matd getProjectionMatrix(const vo::Matd& K, const vo::Matd& pose);

// DLT triangulation of N correspondences seen through P0 and P1 (3 x 4 each): one 4 x 4 SVD per point
// on the device, then the division by w.  Empty input gives an empty result, unequal sizes throw
// vo::Error (cv asserts).  Reference: triangulation.cpp:46-66 (cv::triangulatePoints + points4DToVec).
std::vector<vo::Point3d> triangulate(const std::vector<vo::Point2f>& points0, const std::vector<vo::Point2f>& points1,
                                     const vo::Matd& P0, const vo::Matd& P1);

// Column-wise conversions of a 4 x N homogeneous (divide by the last row) or 3 x N matrix to points.
// Reference: triangulation.cpp:13-22 and :24-32.
std::vector<vo::Point3d> points4DToVec(const vo::Matd& points4d);
std::vector<vo::Point3d> points3DToVec(const vo::Matd& points3d);

// Relative scale between two consecutive pairs: median over random point pairs of the ratio of their
// 3-D distances in frame0 and frame1 (reference: triangulation.cpp:68-126).  The reference draws from a
// process-global std::mt19937 seeded by std::random_device, which makes its trajectories
// irreproducible; here the generator can be seeded (setScaleSeed) and otherwise starts the same way.
float getScale(const VOFrame& frame0, const VOFrame& frame1, size_t min_points, size_t max_points, float max_3d_dist);
void setScaleSeed(uint32_t seed);

#endif
