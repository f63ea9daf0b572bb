// triangulation.h — mirror of src/sfm/triangulation.h:9-20.
#ifndef VO_HOST_TRIANGULATION_H
#define VO_HOST_TRIANGULATION_H

#include <cstdint>
#include <vector>

#include "vo_frame.h"

std::vector<vo::Point3d> points3DToVec(const vo::Matd& points3d);  // triangulation.cpp:24-32
std::vector<vo::Point3d> points4DToVec(const vo::Matd& points4d);  // triangulation.cpp:13-22

// cv::triangulatePoints + points4DToVec (triangulation.cpp:46-66)
std::vector<vo::Point3d> triangulate(const std::vector<vo::Point2f>& points0, const std::vector<vo::Point2f>& points1,
                                     const vo::Matd& P0, const vo::Matd& P1);

vo::Matd getProjectionMatrix(const vo::Matd& K, const vo::Matd& pose);  // triangulation.cpp:34-44

// median distance ratio of random point pairs (triangulation.cpp:68-126).  The reference draws
// from a process-global std::mt19937 seeded by std::random_device, which makes its trajectories
// irreproducible; here the generator is seedable (setScaleSeed) and defaults to random_device too.
float getScale(const VOFrame& frame0, const VOFrame& frame1, size_t min_points, size_t max_points, float max_3d_dist);
void setScaleSeed(uint32_t seed);

#endif
