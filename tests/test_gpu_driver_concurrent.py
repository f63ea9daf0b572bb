"""Several VisualOdometry drivers side by side (vo_host_vo_run_concurrent): frames of one sequence are serial,
sequences are not.  Two schedules of the same work must both give, sequence for sequence, exactly
the trajectory a single driver computes alone (tests/test_gpu_driver.py pins that one to the oracle):
  threads   - S drivers on S host threads, each with its own contexts and HIP streams;
  lock step - the same threads, but their two device calls per frame meet in ONE vo_lk_track_batch /
              vo_essential_pose_batch call (visual_odometry_amd/host/host_batch.h).
Sequences of different lengths (members leave the group while the others go on) and a min_tracked_points that
makes the drivers re-detect at different frames."""
import numpy as np
import pytest

from visual_odometry_amd import host_api, synth

pytestmark = pytest.mark.gpu

W, H = 640, 192
K = synth.KITTI_K.copy()
K[0, 2], K[1, 2] = 320.0, 96.0


def _alone(frames, seed, min_tracked):
    vo = host_api.HostVisualOdometry(K, min_tracked, seed, 3)
    out = np.stack([vo.add_image(f).reshape(12) for f in frames])
    vo.close()
    return out


@pytest.mark.parametrize("min_tracked", [750, 1500])
def test_concurrent_drivers_equal_single_drivers(min_tracked):
    base, _, _ = synth.make_sequence(5, 26, w=W, h=H, K=K)
    seqs = [base[:26], base[3:20], base[::-1][:23].copy(), base[5:9], base[1:25:2].copy()]
    want = [_alone(s, 1234 + i, min_tracked) for i, s in enumerate(seqs)]
    assert any(not np.array_equal(w[-1], np.eye(3, 4).reshape(12)) for w in want)  # the drivers did move
    for lockstep in (0, 1, 2):  # threads only; one lock-step group; two groups
        wall, timed, poses = host_api.run_concurrent(K, seqs, min_tracked_points=min_tracked, scale_seed=1234,
                                                     tracker_channels=3, warmup=2, want_poses=True, lockstep=lockstep)
        assert timed == sum(len(s) - 2 for s in seqs) and wall > 0
        for i, (w, g) in enumerate(zip(want, poses)):
            assert np.array_equal(w, g), (lockstep, i)


def test_lockstep_group_of_one_and_many():
    base, _, _ = synth.make_sequence(6, 12, w=W, h=H, K=K)
    want = _alone(base, 1234, 750)
    _, _, p1 = host_api.run_concurrent(K, [base], want_poses=True, lockstep=True)
    assert np.array_equal(p1[0], want)
    _, _, p16 = host_api.run_concurrent(K, [base] * 16, want_poses=True, lockstep=True, scale_seed=1234)
    # driver s draws getScale's samples from generator 1234 + s: driver 0 is the single driver, the others differ
    # from it only through the scale (their E, R, t are those of driver 0 - the same frames)
    assert np.array_equal(p16[0], want)
    for s in range(1, 16):
        assert np.allclose(p16[s].reshape(-1, 3, 4)[:, :, :3], want.reshape(-1, 3, 4)[:, :, :3], atol=0, rtol=0)
