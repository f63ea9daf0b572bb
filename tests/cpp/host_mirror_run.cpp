// host_mirror_run.cpp — drives the host-side mirror of the reference's interfaces the way the
// reference's own test does (src/sfm/test/triangulation_test.cpp:40-66: detectFAST -> association ->
// updatePose -> triangulate through getProjectionMatrix), with descriptor matching as the
// association step (detectComputeORB + matcher(), the key-frame path of
// src/sfm/bundle_adjustment.cpp:41,58) because the LK tracker is a "next" row.
// Usage: host_mirror_run <frames.raw> <w> <h> <nfeatures> <out_dir>   (two 8-bit gray frames)
// Every intermediate result is dumped as a raw little-endian file for tests/test_gpu_host_mirror.py.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../visual_odometry_amd/host/feature_detector.h"
#include "../../visual_odometry_amd/host/matcher.h"
#include "../../visual_odometry_amd/host/vo_pose.h"

static void dump(const std::string& dir, const char* name, const void* p, size_t bytes) {
    FILE* f = fopen((dir + "/" + name).c_str(), "wb");
    if (!f) {
        perror(name);
        exit(2);
    }
    if (bytes) fwrite(p, 1, bytes, f);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc != 6) {
        fprintf(stderr, "usage: %s frames.raw w h nfeatures out_dir\n", argv[0]);
        return 2;
    }
    const int w = atoi(argv[2]), h = atoi(argv[3]), nfeatures = atoi(argv[4]);
    const std::string out = argv[5];
    FILE* f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 2;
    }
    VOFrame vo0, vo1;
    for (VOFrame* v : {&vo0, &vo1}) {
        vo::Image im;
        im.width = w;
        im.height = h;
        im.px.resize((size_t)w * h);
        if (fread(im.px.data(), 1, im.px.size(), f) != im.px.size()) {
            fprintf(stderr, "short read\n");
            return 2;
        }
        v->setImage(std::move(im));
    }
    fclose(f);
    try {
        vo::Matd K = vo::Matd::eye(3, 3);
        K(0, 0) = 718.856 * w / 1241.0;
        K(1, 1) = 718.856 * w / 1241.0;
        K(0, 2) = 607.1928 * w / 1241.0;
        K(1, 2) = 185.2157 * h / 376.0;

        FeatureDetector feature_detector(nfeatures);
        VOFrame probe = vo0;
        feature_detector.detectFAST(&probe);  // appends kp.pt to probe.points
        dump(out, "fast_points.bin", probe.points.data(), probe.points.size() * sizeof(vo::Point2f));

        std::vector<vo::ImageFeatures> features(2);
        VOFrame* fr[2] = {&vo0, &vo1};
        for (int i = 0; i < 2; i++) {
            features[i].img_idx = i;
            features[i].img_size = {w, h};
            feature_detector.detectComputeORB(*fr[i], &features[i].keypoints, &features[i].descriptors);
            dump(out, i ? "kp1.bin" : "kp0.bin", features[i].keypoints.data(),
                 features[i].keypoints.size() * sizeof(vo::KeyPoint));
            dump(out, i ? "desc1.bin" : "desc0.bin", features[i].descriptors.d.data(), features[i].descriptors.d.size());
        }
        const auto pairwise = matcher(features, K);
        if (pairwise.size() != 1) {
            fprintf(stderr, "matcher returned %zu pairs\n", pairwise.size());
            return 3;
        }
        dump(out, "matches.bin", pairwise[0].matches.data(), pairwise[0].matches.size() * sizeof(vo::DMatch));
        dump(out, "inliers.bin", pairwise[0].inliers_mask.data(), pairwise[0].inliers_mask.size());
        const auto tracks = createMatchMatrix(pairwise, 2);
        const int ntracks = (int)tracks.size();
        dump(out, "ntracks.bin", &ntracks, sizeof(int));

        for (const auto& m : pairwise[0].matches) {
            vo0.points.push_back(features[0].keypoints[m.queryIdx].pt);
            vo1.points.push_back(features[1].keypoints[m.trainIdx].pt);
        }
        setScaleSeed(1234);
        const int rc = updatePose(K, &vo0, &vo1);
        dump(out, "rc.bin", &rc, sizeof(int));
        dump(out, "E.bin", vo1.E.d.data(), 9 * sizeof(double));
        dump(out, "mask.bin", vo1.mask.d.data(), vo1.mask.d.size());
        dump(out, "local_R.bin", vo1.local_R.d.data(), 9 * sizeof(double));
        dump(out, "local_t.bin", vo1.local_t.d.data(), 3 * sizeof(double));
        dump(out, "points_3d.bin", vo1.points_3d.data(), vo1.points_3d.size() * sizeof(vo::Point3d));
        dump(out, "pose.bin", vo1.pose.d.data(), 12 * sizeof(double));

        std::vector<vo::Point2f> p0, p1;
        for (size_t i = 0; i < vo0.points.size(); i++)
            if (vo1.mask.at((int)i)) {
                p0.push_back(vo0.points[i]);
                p1.push_back(vo1.points[i]);
            }
        const auto tri = triangulate(p0, p1, getProjectionMatrix(K, vo0.pose), getProjectionMatrix(K, vo1.pose));
        dump(out, "tri.bin", tri.data(), tri.size() * sizeof(vo::Point3d));
        // AKAZE as BundleAdjustment::addKeyFrame calls it (bundle_adjustment.cpp:41); BRISK is never called by the
        // reference and is not built: it must say so
        {
            std::vector<vo::KeyPoint> k;
            vo::Descriptors d;
            feature_detector.detectComputeAKAZE(vo0, &k, &d);
            if (d.cols != 61 || d.rows != (int)k.size()) return 5;
            dump(out, "akaze_kp.bin", k.data(), k.size() * sizeof(vo::KeyPoint));
            dump(out, "akaze_desc.bin", d.d.data(), d.d.size());
        }
        bool threw = false;
        try {
            std::vector<vo::KeyPoint> k;
            vo::Descriptors d;
            feature_detector.detectComputeBRISK(vo0, &k, &d);
        } catch (const vo::Error&) {
            threw = true;
        }
        if (!threw) return 4;
    } catch (const vo::Error& e) {
        fprintf(stderr, "vo::Error %d: %s\n", e.code, e.what());
        return 1;
    }
    printf("ok\n");
    return 0;
}
