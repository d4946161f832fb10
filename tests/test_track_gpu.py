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


def test_pyr_down_size_sweep_and_unaligned_views(vs, cuda):
    """The kernel splits every row into interior groups of four outputs (16-byte windows, dot products on packed dwords)
    and edge groups (reflected taps, partial stores) that run in workgroups of their own: every width from 5 to 52 with
    heights around the 4-row reflection limit, then views whose base or pitch is not 4-byte aligned (all-edge path)."""
    rng = np.random.default_rng(11)
    for h in (1, 2, 3, 4, 5, 8, 9):
        for w in range(5, 53):
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
            assert np.array_equal(vs.pyr_down(dev(img, cuda)).cpu().numpy(), oracle.pyr_down(img)), (w, h)
    big = rng.integers(0, 256, (70, 203), dtype=np.uint8)
    d = dev(big, cuda)
    for x0 in (0, 1, 2, 3):
        for w in (64, 65, 66, 67, 131):
            v = big[3:64, x0:x0 + w]
            assert np.array_equal(vs.pyr_down(d[3:64, x0:x0 + w]).cpu().numpy(), oracle.pyr_down(np.ascontiguousarray(v))), (x0, w)


def test_pyr_down_two_levels_in_one_launch(vs, cuda):
    """vstab_pyr_down_x2 (levels 2 and 3 of the LK pyramid in one launch): both levels equal two pyrDown calls of the oracle --
    the pipeline's level sizes (1920x1080 and 960x540 are what a 4K / 1080p frame hands it), every residue of the tile sizes,
    odd sizes, images smaller than a tile, the smallest ones (which fall back to two launches), views whose
    base or pitch is not 4-byte aligned."""
    rng = np.random.default_rng(12)
    sizes = [(1920, 1080), (960, 540), (640, 360), (333, 181), (57, 49), (56, 48), (55, 47), (29, 25), (28, 24), (27, 23), (16, 16), (15, 15), (9, 7), (5, 3)]
    sizes += [(w, 33) for w in range(100, 196)] + [(61, h) for h in range(40, 124)]   # every residue of the 16 x 10 tile (64 x 40 source pixels) and of the former 14 x 12
    for w, h in sizes:
        img = synth.luma(w + h, w, h) if w >= 320 else rng.integers(0, 256, (h, w), dtype=np.uint8)
        mid, dst = vs.pyr_down_x2(dev(img, cuda))
        e1 = oracle.pyr_down(img)
        assert np.array_equal(mid.cpu().numpy(), e1), (w, h, "first level")
        assert np.array_equal(dst.cpu().numpy(), oracle.pyr_down(e1)), (w, h, "second level")
    big = rng.integers(0, 256, (90, 203), dtype=np.uint8)
    d = dev(big, cuda)
    for x0 in (0, 1, 2, 3):
        for w in (64, 65, 131):
            v = np.ascontiguousarray(big[3:84, x0:x0 + w])
            mid, dst = vs.pyr_down_x2(d[3:84, x0:x0 + w])
            assert np.array_equal(mid.cpu().numpy(), oracle.pyr_down(v)) and np.array_equal(dst.cpu().numpy(), oracle.pyr_down(oracle.pyr_down(v))), (x0, w)


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


def test_fused_and_two_pass_detectors_agree_with_the_oracle(vs, cuda):
    """The one-pass detector (k_corners_fused + k_filter_keys) and the two-pass one (k_min_eig + k_corner_candidates)
    on ragged sizes around the 64 x 31 tile, pitched views, sizes smaller than one tile, and every corner kept
    (max_corners large, min_distance 0 -> the full candidate list in order, i.e. the threshold and the 3x3 test)."""
    cases = [(11, 64, 31), (12, 65, 32), (13, 63, 30), (14, 129, 63), (15, 200, 95), (16, 640, 360), (17, 9, 7), (18, 3, 3), (19, 70, 3), (20, 3, 70)]
    for seed, w, h in cases:
        g = synth.luma(seed, w, h) if min(w, h) > 30 else np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)
        exp = oracle.good_features(g, 4000, 0.01, 0.0)
        for det in (vs.DETECTOR_AUTO, vs.DETECTOR_TWO_PASS):
            info = {}
            got = vs.good_features(dev(g, cuda), 4000, 0.01, 0.0, detector=det, info=info)
            assert np.array_equal(got, exp), (w, h, det, len(got), len(exp))
            assert info["detector_used"] == (vs.DETECTOR_FUSED if det == vs.DETECTOR_AUTO else vs.DETECTOR_TWO_PASS)
    # pitched view (row pitch != width, base not 4-byte aligned)
    big = synth.luma(21, 400, 200)
    view = dev(big, cuda)[3:150, 5:298]
    exp = oracle.good_features(np.ascontiguousarray(big[3:150, 5:298]), 300, 0.01, 5.0)
    assert np.array_equal(vs.good_features(view, 300, 0.01, 5.0, detector=vs.DETECTOR_AUTO), exp)
    # other quality levels: the lower-bound filter inside the tile uses the same factor
    g = synth.luma(22, 640, 360)
    for q in (0.3, 0.001, 1e-6):
        assert np.array_equal(vs.good_features(dev(g, cuda), 1000, q, 3.0, detector=vs.DETECTOR_AUTO), oracle.good_features(g, 1000, q, 3.0)), q


def test_fused_detector_on_eigenvalue_plateaus(vs, cuda):
    """A tile has 256 key slots.  White noise stays far below (~150 local maxima per 64 x 31 tile).  A 2-px checkerboard
    makes the eigenvalue map one plateau (every pixel a non-strict 3x3 maximum, 1984 per tile): such tiles hand over a
    dense map instead of keys, same corners (ties: later raster position first).  Only when more corners pass the FINAL
    threshold than the key buffer holds (2^18) does the two-pass detector take over, which grows its buffer."""
    rng = np.random.default_rng(5)
    g = rng.integers(100, 110, (1080, 1920), dtype=np.uint8)
    info = {}
    got = vs.good_features(dev(g, cuda), detector=vs.DETECTOR_AUTO, info=info)
    assert info["detector_used"] == vs.DETECTOR_FUSED
    assert np.array_equal(got, oracle.good_features(g)) and len(got) == 200

    def checker(w, h):
        return ((np.add.outer(np.arange(h) // 2, np.arange(w) // 2) % 2) * 100 + 50).astype(np.uint8)
    h, w = 360, 640
    g = checker(w, h)
    for mc, md in ((200, 30.0), (3000, 0.0)):
        got = vs.good_features(dev(g, cuda), mc, 0.01, md, detector=vs.DETECTOR_AUTO, info=info)
        assert info["detector_used"] == vs.DETECTOR_FUSED
        assert np.array_equal(got, oracle.good_features(g, mc, 0.01, md)) and len(got) == mc
    # half the frame plateau, half texture; ragged size: plateau tiles cut by the right and bottom image edges
    g[:, : w // 2] = synth.luma(3, w, h)[:, : w // 2]
    got = vs.good_features(dev(g, cuda), 500, 0.01, 4.0, detector=vs.DETECTOR_AUTO, info=info)
    assert np.array_equal(got, oracle.good_features(g, 500, 0.01, 4.0))
    g = checker(333, 181)
    got = vs.good_features(dev(g, cuda), 2000, 0.01, 0.0, detector=vs.DETECTOR_AUTO, info=info)
    assert info["detector_used"] == vs.DETECTOR_FUSED and np.array_equal(got, oracle.good_features(g, 2000, 0.01, 0.0))
    # 1280 x 720 of plateau: ~9e5 corners above the threshold
    g = checker(1280, 720)
    got = vs.good_features(dev(g, cuda), 300, 0.01, 20.0, detector=vs.DETECTOR_AUTO, info=info)
    assert info["detector_used"] == vs.DETECTOR_TWO_PASS
    assert np.array_equal(got, oracle.good_features(g, 300, 0.01, 20.0)) and len(got) == 300


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


def test_lk_features_leaving_the_image_final_position_rule(vs, cuda):
    """OpenCV's test of the final position behind the iteration loop (the reference passes `err`,
    FrameSourceWarp.cpp:250-259): seeded pairs in which a feature is dropped by that rule alone
    (tests/test_oracle_cpu.py shows it decides); HIP = oracle in every coordinate bit and every status byte."""
    from test_oracle_cpu import LK_EDGE_SEEDS
    dropped = 0
    for seed in LK_EDGE_SEEDS:
        prev, nxt, pts = synth.edge_leaving_pair(seed)
        exp, est = oracle.pyr_lk(prev, nxt, pts)
        got, gst = vs.pyr_lk(dev(prev, cuda), dev(nxt, cuda), pts)
        assert np.array_equal(gst, est), (seed, np.nonzero(gst != est)[0])
        ok = est > 0
        assert np.array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32)), seed
        oracle.set_lk_final_check(False)
        try:
            _, loose = oracle.pyr_lk(prev, nxt, pts)
        finally:
            oracle.set_lk_final_check(True)
        dropped += int((loose != est).sum())
    assert dropped >= len(LK_EDGE_SEEDS)
