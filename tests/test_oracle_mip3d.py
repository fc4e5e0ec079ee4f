"""The depth pass of 3-D mip generation (generateMips3d, lib/src/Texture.cpp:103-227) -- all of its
arithmetic is in the reference tree, so the oracle is checked against known answers derived by
hand from that source."""
import numpy as np

import oracle_lib as O


def _vol(n, h=2, w=3, seed=0):
    rng = np.random.default_rng(seed)
    return rng.random((n, h, w, 4)).astype(np.float32)


def test_box_halving_is_the_mean_of_slice_pairs():
    # n_prev = 2*depth: invScale 2, offset 2, filterScale 0.5; slice i counts when |i + 0.5 - (2d + 1)|/2 <= 0.5
    # -> i in {2d, 2d + 1}: the plain average of the pair in double, rounded to float (:114-165)
    v = _vol(8)
    got = O.mip_depth_pass(v, 4, filter=0)
    want = ((v[0::2].astype(np.float64) + v[1::2].astype(np.float64))/2).astype(np.float32)
    assert np.array_equal(got, want)


def test_tent_weights_of_the_else_branch():
    # Linear (and every non-Box filter): weight max(1 - |i + 0.5 - center|/offset, 0) (:166-225).
    # 4 -> 2 slices: center 1 and 3, offset 2: weights of slices 0..3 for d = 0: 0.75, 0.75, 0.25, 0 (end = 3)
    v = _vol(4, seed=1)
    for filt in (1, 3):
        got = O.mip_depth_pass(v, 2, filter=filt)
        a = v.astype(np.float64)
        d0 = (a[0]*0.75 + a[1]*0.75 + a[2]*0.25)/1.75
        d1 = (a[1]*0.25 + a[2]*0.75 + a[3]*0.75)/1.75
        assert np.array_equal(got[0], d0.astype(np.float32))
        assert np.array_equal(got[1], d1.astype(np.float32))


def test_odd_depth_and_single_slice():
    v = _vol(5, seed=2)
    got = O.mip_depth_pass(v, 2, filter=0)           # 5 -> 2: invScale 2.5
    # d = 0: center 1.25, start 0, end min(int(4.25), 5) = 4; kept when |i + 0.5 - 1.25|/2.5 <= 0.5: i = 0, 1, 2
    a = v.astype(np.float64)
    assert np.array_equal(got[0], ((a[0] + a[1] + a[2])/3).astype(np.float32))
    # d = 1: center 3.75, start int(1.75) = 1, end 5; kept: |i - 3.25| <= 1.25: i = 2, 3, 4
    assert np.array_equal(got[1], ((a[2] + a[3] + a[4])/3).astype(np.float32))
    one = O.mip_depth_pass(v[:1], 1, filter=0)
    assert np.array_equal(one, v[:1])


def test_srgb_textures_average_in_linear_space():
    to_lin, to_srgb = O.color_fns()
    v = _vol(2, h=1, w=1, seed=3)
    got = O.mip_depth_pass(v, 1, filter=0, color_space=1)[0, 0, 0]
    for c in range(3):
        lin = [np.float32(to_lin(float(v[i, 0, 0, c]))) for i in range(2)]
        mean = np.float32((float(lin[0]) + float(lin[1]))/2)
        assert got[c] == np.float32(to_srgb(float(mean)))
    assert got[3] == np.float32((float(v[0, 0, 0, 3]) + float(v[1, 0, 0, 3]))/2)


def test_chain_dimensions_follow_the_reference():
    vol = (np.random.default_rng(4).random((6, 10, 12, 4))*255).astype(np.uint8)
    chain = O.mip_chain3d(vol, 4, filter=0)
    assert [c.shape[:3] for c in chain] == [(6, 10, 12), (3, 5, 6), (1, 2, 3), (1, 1, 1)]
