"""createMatchMatrix of the C++ host mirror (indexed: ordered holder sets per (image, keypoint)) against the
reference's algorithm restated line by line (reference: src/matcher/matcher.cpp:55-86): the whole MATRIX,
row order included, on inputs built to collide (shared keypoints, overwrites that orphan or duplicate
entries).  CPU only: nothing here touches the GPU."""
import numpy as np
import pytest

from visual_odometry_amd import host_api


def reference_match_matrix(pairs, pose_count):
    """for pwm in pairwise_matches: for every inlier match: the FIRST row with row[src] == queryIdx or
    row[dst] == trainIdx takes both values, else a new row of -1 with the two values."""
    mm = []
    for src, dst, q, t, inl in pairs:
        for i in range(len(q)):
            if inl[i] != 1:
                continue
            for row in mm:
                if row[src] == q[i] or row[dst] == t[i]:
                    row[src], row[dst] = int(q[i]), int(t[i])
                    break
            else:
                row = [-1] * pose_count
                row[src], row[dst] = int(q[i]), int(t[i])
                mm.append(row)
    return np.array(mm, np.int32).reshape(-1, pose_count)


def random_pairs(rng, n_img, n_kp, n_matches, inlier_p):
    pairs = []
    for i in range(n_img):
        for j in range(i + 1, min(n_img, i + 3)):  # matcher()'s pairs: (i, i+1), (i, i+2)
            q = rng.integers(0, n_kp, n_matches)
            t = rng.integers(0, n_kp, n_matches)
            inl = (rng.random(n_matches) < inlier_p).astype(np.uint8)
            inl[rng.integers(0, n_matches, 2)] = 2  # cv masks are 0 / 1; anything else is not an inlier
            pairs.append((i, j, q, t, inl))
    return pairs


@pytest.mark.parametrize("n_img,n_kp,n_matches,inlier_p,seed", [
    (4, 40, 60, 0.7, 0), (4, 2000, 900, 0.8, 1), (2, 30, 200, 1.0, 2), (5, 15, 100, 0.5, 3), (3, 500, 0, 0.5, 4),
    (4, 6, 300, 0.9, 5)])
def test_match_matrix_equals_the_reference_algorithm(n_img, n_kp, n_matches, inlier_p, seed):
    rng = np.random.default_rng(seed)
    pairs = random_pairs(rng, n_img, n_kp, max(n_matches, 2), inlier_p) if n_matches else \
        [(0, 1, np.zeros(0, int), np.zeros(0, int), np.zeros(0, np.uint8))]
    want = reference_match_matrix(pairs, n_img)
    got = host_api.create_match_matrix(pairs, n_img)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_match_matrix_overwrite_semantics_by_hand():
    # (0:5 - 1:7), then (0:5 - 1:8) overwrites the track's image-1 entry, then (1:7 - 2:3) finds no holder of 1:7
    # any more and starts a second track: exactly what the reference does
    pairs = [(0, 1, [5, 5], [7, 8], [1, 1]), (1, 2, [7], [3], [1])]
    got = host_api.create_match_matrix(pairs, 3)
    assert got.tolist() == [[5, 8, -1], [-1, 7, 3]]
    with pytest.raises(RuntimeError):
        host_api.create_match_matrix([(0, 3, [1], [1], [1])], 3)
