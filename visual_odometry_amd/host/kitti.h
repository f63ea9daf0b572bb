// kitti.h — mirror of the KITTI odometry formats the reference reads and writes
// (src/kitti/kitti.h:8-12, kitti.cpp:10-49; pose files: src/kitti/evaluate_odometry.cpp:16-32,
// dataset layout: README.md:54-94).  Host I/O only.
#ifndef VO_HOST_KITTI_H
#define VO_HOST_KITTI_H

#include <string>
#include <vector>

#include "vo_host_types.h"

// Line `line_number` of calib.txt ("P0: fx 0 cx 0 0 fy cy 0 0 0 1 0"): the first four characters are
// dropped, the rest split on single spaces into 12 doubles, the left 3x3 block is K (kitti.cpp:10-38).
// Throws vo::Error if the file cannot be opened (the reference aborts with LOG(FATAL)).
vo::Matd loadKittiCalibration(std::string calib_file, int line_number);

// 3x4 pose -> 4x4 with the last row 0 0 0 1 (kitti.cpp:41-49; the reference's Matrix type)
vo::Matd kittiResultMat(const vo::Matd& pose);

// poses/NN.txt: twelve doubles per line, row-major 3x4 camera-to-world (evaluate_odometry.cpp:16-32);
// returned as 4x4 matrices.  savePoses writes the same format (the reference only plots).
std::vector<vo::Matd> loadPoses(const std::string& file_name);
void savePoses(const std::string& file_name, const std::vector<vo::Matd>& poses);

#endif
