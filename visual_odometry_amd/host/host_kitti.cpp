// host_kitti.cpp — KITTI calibration and pose files of the host mirror (reference: src/kitti/kitti.cpp:10-49).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <fstream>
#include <sstream>

#include "host_internal.h"
#include "kitti.h"

// ---- KITTI formats ------------------------------------------------------------------------------
vo::Matd loadKittiCalibration(std::string calib_file, int line_number) {
    std::string line;
    std::ifstream file_stream(calib_file);
    if (!file_stream.is_open()) throw vo::Error(VO_ERR_INVALID, "Unable to open file " + calib_file);
    int count = 0;
    while (std::getline(file_stream, line)) {
        if (count == line_number) break;
        count++;
    }
    line.erase(0, std::min<size_t>(4, line.size()));
    // boost::algorithm::split(elements, line, is_any_of(" ")): empty tokens are kept and std::stod
    // throws on them, as the reference would
    double data[12] = {0};
    size_t pos = 0;
    int i = 0;
    for (;;) {
        const size_t sp = line.find(' ', pos);
        const std::string tok = line.substr(pos, sp == std::string::npos ? std::string::npos : sp - pos);
        if (i >= 12) throw vo::Error(VO_ERR_INVALID, "calibration line holds more than 12 values");
        try {
            data[i++] = std::stod(tok);
        } catch (const std::exception&) {
            throw vo::Error(VO_ERR_INVALID, "calibration line: '" + tok + "' is not a number");
        }
        if (sp == std::string::npos) break;
        pos = sp + 1;
    }
    vo::Matd K(3, 3);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) K(r, c) = data[4 * r + c];
    return K;
}

vo::Matd kittiResultMat(const vo::Matd& pose) {
    vo::Matd m = vo::Matd::eye(4, 4);
    for (int r = 0; r < pose.rows; r++)
        for (int c = 0; c < pose.cols; c++) m(r, c) = pose(r, c);
    return m;
}

// Ground-truth / result trajectories: text, one pose per line, the first three rows of the 4 x 4
// camera-to-world matrix row-major (README.md:54-94 of the reference describes the layout).  Lines that do
// not hold twelve numbers end the file (a trailing blank line is fine).
std::vector<vo::Matd> loadPoses(const std::string& file_name) {
    std::vector<vo::Matd> poses;
    std::ifstream in(file_name);
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream fields(line);
        vo::Matd P = vo::Matd::eye(4, 4);
        int got = 0;
        for (double v; got < 12 && (fields >> v); got++) P(got / 4, got % 4) = v;
        if (got == 0 && line.find_first_not_of(" \t\r") == std::string::npos) continue;  // blank line
        if (got != 12) break;
        poses.push_back(P);
    }
    return poses;
}

void savePoses(const std::string& file_name, const std::vector<vo::Matd>& poses) {
    FILE* fp = fopen(file_name.c_str(), "w");
    if (!fp) throw vo::Error(VO_ERR_INVALID, "Unable to open file " + file_name);
    for (const auto& P : poses) {
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) fprintf(fp, "%.9e%s", P(r, c), (r == 2 && c == 3) ? "\n" : " ");
    }
    fclose(fp);
}

