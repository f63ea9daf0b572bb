// matcher.h — mirror of src/matcher/matcher.h:11-13.
#ifndef VO_HOST_MATCHER_H
#define VO_HOST_MATCHER_H

#include <vector>

#include "vo_host_types.h"

// BF-Hamming knnMatch(k=2) + ratio 0.7 + 200 px gate + findEssentialMat(RANSAC, 0.999, 1.0) for
// image pairs (i, i+1), (i, i+2); pairs with fewer than 5 surviving matches are dropped
// (reference: src/matcher/matcher.cpp:8-53).
std::vector<vo::MatchesInfo> matcher(const std::vector<vo::ImageFeatures>& features, const vo::Matd& K);

// feature tracks across poses (reference: src/matcher/matcher.cpp:55-86); host-side.
std::vector<std::vector<int>> createMatchMatrix(const std::vector<vo::MatchesInfo> pairwise_matches,
                                                const int pose_count);

#endif
