"""Builds the instrumented copy of the library that tools/micro/seltime.py loads: the sources of
visual_odometry_amd/csrc copied to tools/micro/variants/seltime (git-ignored) with wall-clock marks in k_orb_select
(phases of every level of frame 0, rounds of the level-0 block)."""
import glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "visual_odometry_amd/csrc")
DST = os.path.join(ROOT, "tools/micro/variants/seltime")
os.makedirs(DST, exist_ok=True)
for f in glob.glob(SRC + "/*"):
    if os.path.isfile(f) and not f.endswith("Makefile"):
        shutil.copy(f, DST)
mk = open(os.path.join(SRC, "Makefile")).read().replace("../../include/", "../../../../include/").replace(
    "OUT := ../libvo_hip.so", "OUT := ../libvo_hip_seltime.so")
open(os.path.join(DST, "Makefile"), "w").write(mk)
for f in glob.glob(DST + "/*"):
    if f.endswith((".h", ".cpp", ".hip")):
        t = open(f).read().replace('"../../include/', '"../../../../include/')
        open(f, "w").write(t)
p = os.path.join(DST, "k_features.hip")
s = open(p).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:60])
    s = s.replace(old, new)


rep("namespace {\n\nstruct RI {", """__device__ unsigned long long g_sel_t[8 * 8];
__device__ unsigned long long g_rounds[512];
__device__ int g_nrounds;
#define SELT(k) if (tid == 0 && f == 0) g_sel_t[l * 8 + (k)] = wall_clock64();
__device__ unsigned long long g_ld_t[8 * 4];
#define LDT(k) if (tid == 0 && f == 0) g_ld_t[l * 4 + (k)] = wall_clock64();
extern "C" int vo_debug_ld_times(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ld_t), sizeof(g_ld_t)); }
#define RLOG(tag, range) if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && g_nrounds < 500) { g_rounds[g_nrounds++] = ((unsigned long long)(tag) << 60) | ((unsigned long long)(range) << 40) | (wall_clock64() & 0xffffffffffull); }
extern "C" int vo_debug_sel_times(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sel_t), sizeof(g_sel_t)); }
extern "C" int vo_debug_sel_rounds(unsigned long long* out, int* n) { (void)hipMemcpyFromSymbol(n, HIP_SYMBOL(g_nrounds), 4); return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rounds), sizeof(g_rounds)); }
namespace {

struct RI {""")
rep("        if (lane == 0) sh[wv] = sum;\n        __syncthreads();\n        int run = 0;", "        LDT(0)\n        if (lane == 0) sh[wv] = sum;\n        __syncthreads();\n        LDT(1)\n        int run = 0;")
rep("    __syncthreads();\n    {\n        const uint8_t* S = level_px(score, g, f, l);", "    LDT(2)\n    __syncthreads();\n    LDT(3)\n    {\n        const uint8_t* S = level_px(score, g, f, l);")
rep("    const int wv = tid >> 6, lane = tid & 63;\n    const int seg0", "    const int wv = tid >> 6, lane = tid & 63;\n    SELT(0)\n    if (tid == 0 && f == 0 && l == 0) g_nrounds = 0;\n    const int seg0")
rep("    __syncthreads();\n    n = block_retain_best(v, n, 2 * quota.q[l], Lp, Ra, sh, tid);",
    "    __syncthreads();\n    if (tid == 0 && f == 0) g_sel_t[l * 8 + 7] = n;\n    SELT(1)\n    n = block_retain_best(v, n, 2 * quota.q[l], Lp, Ra, sh, tid);\n    SELT(2)")
rep("    __syncthreads();\n    n = block_retain_best(v, n, quota.q[l], Lp, Ra, sh, tid);",
    "    __syncthreads();\n    SELT(3)\n    n = block_retain_best(v, n, quota.q[l], Lp, Ra, sh, tid);\n    SELT(4)")
rep("    if (tid == 0) *cnt = n;\n}\n\n// keypoints per frame", "    if (tid == 0) *cnt = n;\n    SELT(5)\n}\n\n// keypoints per frame")
rep("    while (last - first > 64) {\n        if (depth == 0) return false;", "    while (last - first > 64) {\n        RLOG(1, last - first)\n        if (depth == 0) return false;")
rep("    if (last - first > 3) {\n        if (tid < 64) {", "    RLOG(2, last - first)\n    if (last - first > 3) {\n        if (tid < 64) {")
rep("    const float ambiguous = v[n_points - 1].r;\n    return block_partition<1>(v, n_points, n, ambiguous, Lp, Ra, sh, tid);",
    "    const float ambiguous = v[n_points - 1].r;\n    RLOG(4, n - n_points)\n    const int rr = block_partition<1>(v, n_points, n, ambiguous, Lp, Ra, sh, tid);\n    RLOG(5, rr)\n    return rr;")
open(p, "w").write(s)
sys.exit(subprocess.call(["make", "-C", DST, "-j8"], stdout=subprocess.DEVNULL))
