"""GPU parity tests of the tracking front-end (a3, a4) through the C ABI: bit-exact corner
indices, bit-exact pyramid bytes and LK tracks against the CPU oracle and the golden vectors."""
import os

import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def dev(a, cuda):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def test_pyr_down_bit_exact(vs, cuda):
    for seed, w, h in [(1, 320, 180), (2, 333, 181), (3, 1920, 1080), (4, 23, 5)]:
        img = synth.luma(seed, w, h) if w > 30 else np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)
        assert np.array_equal(vs.pyr_down(dev(img, cuda)).cpu().numpy(), oracle.pyr_down(img)), (w, h)
    # pitched source view
    big = np.random.default_rng(5).integers(0, 256, (64, 100), dtype=np.uint8)
    assert np.array_equal(vs.pyr_down(dev(big, cuda)[:, :77]).cpu().numpy(), oracle.pyr_down(big[:, :77]))


def test_min_eig_bit_exact(vs, cuda):
    for seed, w, h in [(1, 320, 180), (2, 333, 181), (6, 64, 48)]:
        img = synth.luma(seed, w, h)
        got = vs.min_eig(dev(img, cuda)).cpu().numpy()
        exp = oracle.min_eig(img)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (w, h, np.abs(got - exp).max())


def test_good_features_bit_exact_indices(vs, cuda):
    kat = np.load(os.path.join(GOLD, "oracle_kat.npz"))
    seed, w, h = (int(v) for v in kat["gftt_seed"])
    g = synth.luma(seed, w, h)
    assert np.array_equal(vs.good_features(dev(g, cuda)), kat["gftt_corners"])
    for seed, w, h, mc, md in [(7, 640, 360, 200, 30.0), (8, 1920, 1080, 200, 30.0), (9, 333, 181, 50, 10.0), (10, 320, 180, 500, 0.0)]:
        g = synth.luma(seed, w, h)
        got = vs.good_features(dev(g, cuda), mc, 0.01, md)
        exp = oracle.good_features(g, mc, 0.01, md)
        assert np.array_equal(got, exp), (w, h, len(got), len(exp))
        assert len(got) > 10


def test_good_features_flat_image_has_no_corners(vs, cuda):
    g = np.full((60, 80), 93, np.uint8)
    assert len(vs.good_features(dev(g, cuda))) == 0 and len(oracle.good_features(g)) == 0


def test_pyr_lk_bit_exact(vs, cuda):
    kat = np.load(os.path.join(GOLD, "oracle_kat.npz"))
    seed, w, h = (int(v) for v in kat["gftt_seed"])
    g0 = synth.luma(seed, w, h)
    g1 = synth.shifted(g0, *kat["lk_shift"])
    nxt, st = vs.pyr_lk(dev(g0, cuda), dev(g1, cuda), kat["gftt_corners"])
    assert np.array_equal(st, kat["lk_status"])
    assert np.array_equal(nxt.view(np.uint32), kat["lk_next"].view(np.uint32))


@pytest.mark.parametrize("w,h,shift", [(640, 360, (3.3, -1.2)), (1920, 1080, (-7.6, 4.1)), (200, 120, (0.4, 0.7))])
def test_pyr_lk_vs_oracle_incl_border_points(vs, cuda, w, h, shift):
    g0 = synth.luma(31, w, h)
    g1 = synth.shifted(g0, *shift)
    pts = oracle.good_features(g0, 200, 0.01, 15.0)
    # add points on / beyond the border: status 0 paths and REFLECT_101 window reads
    extra = np.array([[0, 0], [w - 1, h - 1], [2.5, h - 3.25], [w - 2, 5], [-30, 10], [w + 40, h + 40], [w / 2, h / 2]], np.float32)
    pts = np.concatenate([pts, extra])
    got, gst = vs.pyr_lk(dev(g0, cuda), dev(g1, cuda), pts)
    exp, est = oracle.pyr_lk(g0, g1, pts)
    assert np.array_equal(gst, est)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), int((got != exp).sum())
    ok = est[:-len(extra)] > 0
    assert ok.sum() > 5
    flow = (exp - pts)[:-len(extra)][ok]
    assert np.abs(np.median(flow, axis=0) - shift).max() < 0.1


def test_pyr_lk_empty_and_tiny(vs, cuda):
    g = synth.luma(1, 64, 48)
    nxt, st = vs.pyr_lk(dev(g, cuda), dev(g, cuda), np.zeros((0, 2), np.float32))
    assert nxt.shape == (0, 2) and st.shape == (0,)
    # 40x30: the pyramid stops after level 0 (next level would be <= winSize), SURVEY.md A.3
    g = synth.luma(2, 40, 30)
    assert oracle.lib().vo_pyramid_levels(40, 30) == 1
    pts = np.array([[20, 15], [10, 10]], np.float32)
    got, gst = vs.pyr_lk(dev(g, cuda), dev(g, cuda), pts)
    exp, est = oracle.pyr_lk(g, g, pts)
    assert np.array_equal(gst, est) and np.array_equal(got, exp)


def test_tracking_parity_at_4k(vs, cuda):
    """BASELINE.json's full size: corner indices and LK tracks of a 3840x2160 frame pair, bit for bit."""
    w, h = 3840, 2160
    g0 = synth.luma(41, w, h, rects=400)
    g1 = synth.shifted(g0, 2.6, -1.9)
    got = vs.good_features(dev(g0, cuda))
    exp = oracle.good_features(g0)
    assert np.array_equal(got, exp) and len(got) == 200
    nxt, st = vs.pyr_lk(dev(g0, cuda), dev(g1, cuda), exp)
    onxt, ost = oracle.pyr_lk(g0, g1, exp)
    assert np.array_equal(st, ost) and np.array_equal(nxt.view(np.uint32), onxt.view(np.uint32))
    assert st.sum() > 150
