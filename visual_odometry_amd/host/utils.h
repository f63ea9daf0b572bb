// utils.h — mirror of src/features/utils.h:9-58.
#ifndef VO_HOST_UTILS_H
#define VO_HOST_UTILS_H

#include <vector>

#include "vo_frame.h"

// Drops the points whose status is 0, whose new position is negative or whose displacement is more
// than two standard deviations from the mean, from both frames, and the same rows of vo0's mask
// and points_3d.  Returns what the reference returns: the middle element of its displacement list
// after the removals, truncated to int.
int removePoints(VOFrame* vo0, VOFrame* vo1, std::vector<unsigned char>* status);

#endif
