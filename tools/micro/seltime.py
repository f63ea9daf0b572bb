"""Phase times of k_orb_select for frame 0 (instrumented variant build, tools/micro/variants/seltime)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["VO_HIP_LIB"] = os.path.join(ROOT, "tools/micro/variants/libvo_hip_seltime.so")
import numpy as np, torch
from visual_odometry_amd import api, synth, _lib
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 49
frames, _, K = synth.make_sequence(0, nfr, workers=8)
d = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
ctx = api.VoContext(0)
prm = api.pair_params(2000)
for _ in range(3):
    ctx.pairs_run_dev(d.data_ptr(), nfr, 1241, 376, 1241, 1241 * 376, K, prm)
t = (C.c_ulonglong * 64)()
_lib.lib().vo_debug_sel_times(t)
names = ["load", "retain1", "harris", "retain2", "writeout"]
for l in range(8):
    r = [t[l * 8 + k] for k in range(8)]
    print("level %d n=%5d start+%6.1f us: " % (l, r[7], (r[0] - t[0]) / 100.0) +
          "  ".join("%s %5.1f" % (names[k], (r[k + 1] - r[k]) / 100.0) for k in range(5)) + "  total %.1f us" % ((r[5] - r[0]) / 100.0))

ld = (C.c_ulonglong * 32)()
_lib.lib().vo_debug_ld_times(ld)
for l in range(8):
    print("level %d list build: masks+count %5.2f  barrier %5.2f  scatter %5.2f  barrier %5.2f  scores %5.2f us" % (
        l, (ld[l * 4] - t[l * 8]) / 100.0, (ld[l * 4 + 1] - ld[l * 4]) / 100.0, (ld[l * 4 + 2] - ld[l * 4 + 1]) / 100.0,
        (ld[l * 4 + 3] - ld[l * 4 + 2]) / 100.0, (t[l * 8 + 1] - ld[l * 4 + 3]) / 100.0))
r = (C.c_ulonglong * 512)()
n = C.c_int(0)
_lib.lib().vo_debug_sel_rounds(r, C.byref(n))
prev = None
tags = {1: "block round", 2: "wave rounds", 3: "insertion", 4: "partition", 5: "done"}
for k in range(n.value):
    tag, rng, t = r[k] >> 60, (r[k] >> 40) & 0xfffff, r[k] & 0xffffffffff
    print("%-12s range %5d  +%5.2f us" % (tags.get(tag, "?"), rng, 0.0 if prev is None else (t - prev) / 100.0))
    prev = t
