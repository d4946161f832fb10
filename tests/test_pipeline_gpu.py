"""GPU tests of the pipeline object (vstab_create / vstab_pull_frame) against the oracle's
restatement of FrameSourceWarp::consume_frame / pull_frame (FrameSourceWarp.cpp:397-476)."""
import os

import numpy as np
import pytest

import expect
import oracle
import synth

pytestmark = pytest.mark.gpu

W, H, R_SMOOTH, N = 640, 360, 5, 40


@pytest.fixture(scope="module")
def clip():
    K = oracle.get_preset_camera(4, W, H)
    frames, rots = synth.shaky_clip(3, K, W, H, N, sigma=0.004)
    return K, frames, rots


def run_product(vs, cuda, frames, **cfg):
    import torch
    dev_frames = [torch.from_numpy(f).to(cuda) for f in frames]
    stab = vs.Stabilizer(dev_frames, total=len(frames), **cfg)
    outs = []
    while True:
        o = stab.pull()
        if o is None:
            break
        outs.append(o.cpu().numpy())
    return stab, outs


@pytest.mark.parametrize("prec", [expect.OPENCL, expect.IEEE])
def test_pipeline_matches_oracle_state_machine(vs, cuda, clip, prec):
    K, frames, rots = clip
    stab, outs = run_product(vs, cuda, frames, smooth_radius=R_SMOOTH, seed=11, map_precision=prec)
    assert len(outs) == N - 1                                   # first frame never emitted (:403-407)
    log = stab.frame_log()
    assert len(log) == N - 1
    Ko, (cw, ch) = oracle.get_output_camera(K, W, H)
    assert stab.out_size == (cw, ch) and np.allclose(stab.K_out, Ko, atol=1e-11)

    # oracle state machine: oracle corner detector, oracle LK and the oracle's own restatement of guess_camera_rotation
    # (oracle/geometry.py: same seeded PCG32 stream as vstab_config.seed, nothing taken from the product's log)
    counts = []
    rng = oracle.Pcg32(11)

    def track(prev, cur, corners):
        nxt, st = oracle.pyr_lk(prev, cur, corners)
        counts.append((len(corners), int((st > 0).sum())))
        return corners[st > 0], nxt[st > 0]

    def estimate(pp, cp):
        return oracle.estimate_rotation(pp, cp, K, Ko, rng)

    warp_rots = []
    sm = oracle.WarpStateMachine(frames, R_SMOOTH, lambda g: oracle.good_features(np.ascontiguousarray(g)), track, estimate,
                                 lambda f, R: (warp_rots.append(R), f)[1])
    exp_frames = []
    while True:
        o = sm.pull_frame()
        if o is None:
            break
        exp_frames.append(o)
    assert len(exp_frames) == N - 1
    # key-frame decisions, corner counts and tracked counts are identical (bit-exact detector + tracker)
    assert [l["key"] for l in log] == [l["key"] for l in sm.log]
    assert [(l["n_corners"], l["n_tracked"]) for l in log] == counts
    # rotation estimates: the product's RANSAC + refit against the oracle's independent restatement, frame by frame
    assert [l["inliers"] for l in log] == [l["inliers"] for l in sm.log]
    for a, b in zip(log, sm.log):
        assert np.allclose(a["R"], b["R"], atol=1e-9)
    # smoothing: rotation handed to the warp agrees with the numpy SG filter / inverses
    for i in range(N - 1):
        assert np.allclose(stab.warp_rotation(i), warp_rots[i], atol=1e-9), i
    # pixels: bit-exact against the oracle warp of the same frame with the product's own rotation
    for i in [0, 1, R_SMOOTH, N - 2]:
        p = oracle.map_params(K, Ko, stab.warp_rotation(i))
        assert np.array_equal(outs[i], expect.warp(exp_frames[i], p, cw, ch, prec)), i


def test_rotation_estimates_follow_ground_truth_and_stabilise(vs, cuda, clip):
    K, frames, rots = clip
    stab, outs = run_product(vs, cuda, frames, smooth_radius=R_SMOOTH, seed=5)
    log = stab.frame_log()
    errs = []
    for k, lg in enumerate(log, start=1):
        true_delta = rots[k] @ rots[k - 1].T                   # d_cam_k = R_k R_{k-1}^T d_cam_{k-1}
        errs.append(oracle.rotation_angle(lg["R"] @ true_delta.T))
        assert lg["inliers"] >= 40 and not lg["fallback"]
    assert np.median(errs) < 1.5e-3 and max(errs) < 6e-3, (np.median(errs), max(errs))
    acc_err = oracle.rotation_angle(log[-1]["R_accum"] @ (rots[-1] @ rots[0].T).T)
    assert acc_err < 0.02


@pytest.mark.parametrize("prec", [expect.OPENCL, expect.IEEE])
def test_undistort_only_mode_is_identity_warp(vs, cuda, clip, prec):
    """BASELINE config 1: tracking off -> every emitted frame is the plain undistortion (from the third frame on it is warped
    from the quantised map written once: the CACHED kernel, in the handle's map precision)."""
    K, frames, _ = clip
    stab, outs = run_product(vs, cuda, frames[:8], smooth_radius=2, tracking=0, map_precision=prec)
    assert len(outs) == 7
    Ko, (cw, ch) = oracle.get_output_camera(K, W, H)
    p = oracle.map_params(K, Ko, np.eye(3))
    for i in (0, 1, 2, 6):
        assert np.allclose(stab.warp_rotation(i), np.eye(3), atol=1e-12)
        assert np.array_equal(outs[i], expect.warp(frames[i + 1], p, cw, ch, prec))


@pytest.mark.parametrize("prec", [expect.OPENCL, expect.IEEE])
def test_nearest_interpolation_through_the_pipeline(vs, cuda, clip, prec):
    """vstab_config.interpolation 0 = INTER_NEAREST: same rotations, nearest-neighbour frames; other flags are refused."""
    K, frames, _ = clip
    stab, outs = run_product(vs, cuda, frames[:8], smooth_radius=2, tracking=0, interpolation=0, map_precision=prec)
    Ko, (cw, ch) = oracle.get_output_camera(K, W, H)
    p = oracle.map_params(K, Ko, np.eye(3))
    assert len(outs) == 7
    for i in (0, 6):
        assert np.array_equal(outs[i], expect.warp(frames[i + 1], p, cw, ch, prec, nearest=True))
    ref, _ = run_product(vs, cuda, frames[:10], smooth_radius=2), None
    near, nouts = run_product(vs, cuda, frames[:10], smooth_radius=2, interpolation=0)
    for i in range(len(nouts)):
        assert np.array_equal(near.warp_rotation(i), ref[0].warp_rotation(i))
    import torch
    one = [torch.from_numpy(frames[0]).to(cuda)]
    with pytest.raises(vs.VstabError):
        vs.Stabilizer(one, total=1, interpolation=2)
    with pytest.raises(vs.VstabError):
        vs.Stabilizer(one, total=1, interpolation=0, lens_mode=1)


def test_pull_frames_is_the_frame_loop_in_one_call(vs, cuda, clip):
    """vstab_pull_frames (the consumer's loop of DisplayImage.cpp:60-70 on the C side) == vstab_pull_frame called n times:
    same frames in the same ring slots, the count at end of stream, EOF afterwards."""
    import torch
    K, frames, _ = clip
    ref, outs = run_product(vs, cuda, frames[:12], smooth_radius=2)
    stab = vs.Stabilizer([torch.from_numpy(f).to(cuda) for f in frames[:12]], total=12, smooth_radius=2)
    cw, ch = stab.out_size
    ring = [torch.zeros((ch, cw, 3), dtype=torch.uint8, device=cuda) for _ in range(4)]
    assert stab.pull_frames_into(ring, 0, 3) == 3
    for i in range(3):
        assert np.array_equal(ring[i].cpu().numpy(), outs[i])
    assert stab.pull_frames_into(ring, 3, 4) == 4          # wraps: frames 3..6 into slots 3, 0, 1, 2
    for i in range(3, 7):
        assert np.array_equal(ring[i % 4].cpu().numpy(), outs[i])
    assert stab.pull_frames_into(ring, 7, 100) == len(outs) - 7   # runs into the end of the stream
    assert np.array_equal(ring[(len(outs) - 1) % 4].cpu().numpy(), outs[-1])
    assert stab.pull_frames_into(ring, 0, 1) == 0
    for i in range(len(outs)):
        assert np.array_equal(stab.warp_rotation(i), ref.warp_rotation(i))


def test_python_callback_source_eof_and_errors(vs, cuda, clip):
    import torch
    K, frames, _ = clip
    gen = (torch.from_numpy(f).to(cuda) for f in frames[:6])
    stab = vs.Stabilizer(gen, smooth_radius=3, seed=1)
    n = 0
    while stab.pull() is not None:
        n += 1
    assert n == 5
    assert stab.pull() is None                                   # EOF is sticky (:465-467)
    with pytest.raises(vs.VstabError):
        vs.Stabilizer(iter([]), smooth_radius=3)                 # no first frame to peek (:214)


def test_short_clip_shorter_than_lookahead(vs, cuda, clip):
    """EOF before the queue fills (:456-461): the drain still emits every buffered frame."""
    K, frames, _ = clip
    stab, outs = run_product(vs, cuda, frames[:4], smooth_radius=30, seed=2)
    assert len(outs) == 3


def test_kalman_and_none_smoothers_run(vs, cuda, clip):
    K, frames, _ = clip
    for sm in (vs.SMOOTHER_KALMAN, vs.SMOOTHER_NONE):
        stab, outs = run_product(vs, cuda, frames[:10], smooth_radius=2, smoother=sm, seed=3)
        assert len(outs) == 9
        if sm == vs.SMOOTHER_NONE:
            assert np.allclose(stab.warp_rotation(4), np.eye(3), atol=1e-9)   # corrected == measured -> no correction


def test_host_memory_frames_and_pitched_planes(vs, cuda, clip):
    """vstab_frame.mem = 1 (host planes, any pitch) must give the same stream as device frames."""
    import ctypes
    K, frames, _ = clip
    n = 10
    ref_stab, ref_outs = run_product(vs, cuda, frames[:n], smooth_radius=3, seed=9)
    # host source with padded pitches, driven through the raw C ABI callbacks
    pitch = W + 32
    bufs = []
    for f in frames[:n]:
        yb = np.zeros((H, pitch), np.uint8)
        ub = np.zeros((H // 2, pitch), np.uint8)
        yb[:, :W], ub[:, :W] = f[:H], f[H:]
        bufs.append((yb, ub))
    state = {"i": 0}

    def fill(out, advance):
        if state["i"] >= n:
            return vs.EOF
        yb, ub = bufs[state["i"]]
        o = out.contents
        o.y, o.uv = yb.ctypes.data, ub.ctypes.data
        o.pitch_y = o.pitch_uv = pitch
        o.width, o.height, o.mem, o.pts = W, H, 1, state["i"]
        if advance:
            state["i"] += 1
        return 0
    pull = vs.PULL_FN(lambda u, o: fill(o, True))
    peek = vs.PULL_FN(lambda u, o: fill(o, False))
    src = vs.Source(pull, peek, None)
    cfg = vs.default_config(smooth_radius=3, seed=9)
    h = ctypes.c_void_p()
    assert vs.lib.vstab_create(ctypes.byref(cfg), ctypes.byref(src), ctypes.byref(h)) == vs.OK
    import torch
    cw, ch = ref_stab.out_size
    outs = []
    while True:
        o = torch.empty((ch, cw, 3), dtype=torch.uint8, device=cuda)
        st = vs.lib.vstab_pull_frame(h, o.data_ptr(), o.stride(0))
        if st == vs.EOF:
            break
        assert st == vs.OK, vs.lib.vstab_last_error()
        outs.append(o.cpu().numpy())
    vs.lib.vstab_destroy(h)
    assert len(outs) == len(ref_outs) == n - 1
    for a, b in zip(outs, ref_outs):
        assert np.array_equal(a, b)


def test_upstream_error_code_is_propagated(vs, cuda, clip):
    """A non-EOF upstream failure (the reference rethrows the int, FrameSourceWarp.cpp:462) -> VSTAB_ERR_SOURCE."""
    import ctypes
    import torch
    K, frames, _ = clip
    dev_frames = [torch.from_numpy(f).to(cuda) for f in frames[:3]]
    state = {"i": 0}

    def fill(out, advance):
        if state["i"] >= 3:
            return 7          # some upstream error code
        f = dev_frames[state["i"]]
        o = out.contents
        o.y, o.uv = f.data_ptr(), f.data_ptr() + H * f.stride(0)
        o.pitch_y = o.pitch_uv = f.stride(0)
        o.width, o.height, o.mem, o.pts = W, H, 0, 0
        if advance:
            state["i"] += 1
        return 0
    pull = vs.PULL_FN(lambda u, o: fill(o, True))
    peek = vs.PULL_FN(lambda u, o: fill(o, False))
    src = vs.Source(pull, peek, None)
    cfg = vs.default_config(smooth_radius=5)
    h = ctypes.c_void_p()
    assert vs.lib.vstab_create(ctypes.byref(cfg), ctypes.byref(src), ctypes.byref(h)) == vs.OK
    o = torch.empty((400, 700, 3), dtype=torch.uint8, device=cuda)
    assert vs.lib.vstab_pull_frame(h, o.data_ptr(), o.stride(0)) == vs.ERR_SOURCE
    assert b"7" in vs.lib.vstab_last_error()
    vs.lib.vstab_destroy(h)


def test_profile_counts_and_stage_times(vs, cuda, clip):
    K, frames, _ = clip
    import torch
    dev_frames = [torch.from_numpy(f).to(cuda) for f in frames[:12]]
    stab = vs.Stabilizer(dev_frames, total=12, smooth_radius=2, seed=4)
    stab.enable_profiling(2)
    n = 0
    while stab.pull() is not None:
        n += 1
    p = stab.profile()
    assert n == 11 and p["frames_emitted"] == 11 and p["frames_consumed"] == 12 and p["warp_launches"] == 11
    assert p["warp_timed"] == 11                                   # level 2 times every launch, level 1 every 8th
    assert p["gpu_warp_ms"] > 0 and p["gpu_lk_ms"] > 0 and p["gpu_pyramid_ms"] > 0 and p["host_estimate_ms"] > 0
    assert p["key_frames"] >= 1


def test_pipeline_at_1080p_baseline_config(vs, cuda):
    """BASELINE config 2 geometry (1920x1080, WIDE169_MEASURED -> 1759x998): a short clip through the whole
    pipeline, every emitted frame bit-exact against the oracle warp with the product's rotation, rotations
    close to ground truth, decisions equal to the oracle state machine."""
    w, h, r, n = 1920, 1080, 2, 6
    K = oracle.get_preset_camera(4, w, h)
    frames, rots = synth.shaky_clip(7, K, w, h, n, sigma=0.003)
    stab, outs = run_product(vs, cuda, frames, smooth_radius=r, seed=2)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    assert (cw, ch) == (1759, 998) and stab.out_size == (cw, ch) and len(outs) == n - 1
    log = stab.frame_log()
    for k, lg in enumerate(log, start=1):
        assert lg["inliers"] >= 40
        assert oracle.rotation_angle(lg["R"] @ (rots[k] @ rots[k - 1].T).T) < 3e-3
    # the same tracking decisions as the oracle (bit-exact detector + tracker at this size)
    corners = oracle.good_features(np.ascontiguousarray(frames[0][:h]))
    nxt, st = oracle.pyr_lk(frames[0][:h], frames[1][:h], corners)
    assert log[0]["n_corners"] == len(corners) and log[0]["n_tracked"] == int((st > 0).sum())
    for i in range(n - 1):
        p = oracle.map_params(K, Ko, stab.warp_rotation(i))
        assert np.array_equal(outs[i], expect.warp(frames[i + 1], p, cw, ch)), i


def _run_raw_device_source(vs, cuda, frames, hold, recycle, p010=False, pool=0, lag=False, **cfg_kw):
    """Drive the raw C ABI with a device-frame source.  recycle: upstream owns ONE surface and overwrites it with
    the next frame inside every pull callback (a decoder recycling its output surface).  pool = P: upstream owns P
    surfaces, hands them out round-robin and rewrites surface i % P (on its own stream) inside the callback of frame i
    -- a decoder's frame pool, hold = P - 1.  lag: the caller's stream is kept busy before every pull so that the warps
    run long after the host has moved on."""
    import ctypes
    import torch
    n = len(frames)
    if p010:   # 10 significant bits at the top of 16-bit samples, low 6 bits filled with junk that must be ignored
        rng = np.random.default_rng(0)
        wide = [((f.astype(np.uint16) << 8) | rng.integers(0, 256, f.shape, dtype=np.uint16)).view(np.int16) for f in frames]
        dev = [torch.from_numpy(x).to(cuda) for x in wide]
    else:
        dev = [torch.from_numpy(f).to(cuda) for f in frames]
    surface = torch.empty_like(dev[0])
    surfaces = [torch.empty_like(dev[0]) for _ in range(pool)]
    side = torch.cuda.Stream() if pool else None
    ballast = torch.randn(4096, 4096, device=cuda) if lag else None
    esz = 2 if p010 else 1
    state = {"i": 0, "loaded": -1}

    def fill(out, advance):
        i = state["i"]
        if i >= n:
            return vs.EOF
        if recycle:
            if state["loaded"] != i:
                surface.copy_(dev[i])                       # rewrites the memory handed out by the previous callback
                torch.cuda.synchronize()
                state["loaded"] = i
            t = surface
        elif pool:
            t = surfaces[i % pool]
            if state["loaded"] != i:
                with torch.cuda.stream(side):               # upstream's own stream: nothing orders it behind the warps
                    t.copy_(dev[i])
                side.synchronize()
                state["loaded"] = i
        else:
            t = dev[i]
        o = out.contents
        o.y, o.uv = t.data_ptr(), t.data_ptr() + H * t.stride(0) * esz
        o.pitch_y = o.pitch_uv = t.stride(0) * esz
        o.width, o.height, o.mem, o.pts, o.hold, o.bit_depth = W, H, 0, i, hold, (10 if p010 else 8)
        if advance:
            state["i"] += 1
        return 0
    pull = vs.PULL_FN(lambda u, o: fill(o, True))
    peek = vs.PULL_FN(lambda u, o: fill(o, False))
    src = vs.Source(pull, peek, None)
    cfg = vs.default_config(**cfg_kw)
    h = ctypes.c_void_p()
    assert vs.lib.vstab_create(ctypes.byref(cfg), ctypes.byref(src), ctypes.byref(h)) == vs.OK
    ow, oh = ctypes.c_int(), ctypes.c_int()
    assert vs.lib.vstab_get_output_info(h, ctypes.byref(ow), ctypes.byref(oh), None, None) == vs.OK
    outs = []
    while True:
        o = torch.empty((oh.value, ow.value, 3), dtype=torch.uint8, device=cuda)
        if lag:
            for _ in range(4):
                ballast @ ballast                            # (cfg.stream is the null stream, torch's current one)
        st = vs.lib.vstab_pull_frame(h, o.data_ptr(), o.stride(0))
        if st == vs.EOF:
            break
        assert st == vs.OK, vs.lib.vstab_last_error()
        outs.append(o if lag else o.cpu().numpy())
    outs = [o.cpu().numpy() if lag else o for o in outs]
    vs.lib.vstab_destroy(h)
    return outs


def _hip_runtime():
    """The HIP runtime this process already uses (PyTorch's copy), for the two calls the test needs beside torch."""
    import ctypes
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    pytest.fail("no HIP runtime mapped")


class _DmaBufPool:
    """Device buffers exported as DMA-BUF fds (hipMemGetHandleForAddressRange): what a decoder's surface pool looks like to the
    library.  Planes sit at byte offset 64 of each object."""

    def __init__(self, frames):
        import ctypes
        self.ct = ctypes
        hip = self.hip = _hip_runtime()
        hip.hipMemGetHandleForAddressRange.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_ulonglong]
        hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipFree.argtypes = [ctypes.c_void_p]
        hip.hipGetErrorString.restype = ctypes.c_char_p
        hip.hipGetErrorString.argtypes = [ctypes.c_int]
        self.size = (frames[0].nbytes + 64 + (1 << 21) - 1) & ~((1 << 21) - 1)        # whole 2 MiB pages
        self.bufs, self.fds = [], []
        for f in frames:
            p = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(p), self.size) == 0
            self.bufs.append(p)
            host = np.ascontiguousarray(f)
            assert hip.hipMemcpy(ctypes.c_void_p(p.value + 64), host.ctypes.data_as(ctypes.c_void_p), host.nbytes, 1) == 0
            fd = ctypes.c_int(-1)
            e = hip.hipMemGetHandleForAddressRange(ctypes.byref(fd), p, self.size, 1, 0)   # hipMemRangeHandleTypeDmaBufFd
            if e != 0:
                self.close()
                pytest.skip("this box cannot export device memory as a DMA-BUF: " + hip.hipGetErrorString(e).decode())
            self.fds.append(fd.value)

    def close(self):
        import torch
        torch.cuda.synchronize()
        for fd in self.fds:
            os.close(fd)
        for p in self.bufs:
            self.hip.hipFree(p)
        self.fds, self.bufs = [], []


def _run_dmabuf_source(vs, cuda, fill, **cfg_kw):
    """Drive a handle from a callback that fills vstab_frame itself; -> (emitted frames, status of the pull that ended the
    stream, its message, the handle's profile counters)."""
    import ctypes
    import torch
    pull, peek = vs.PULL_FN(lambda u, o: fill(o.contents, True)), vs.PULL_FN(lambda u, o: fill(o.contents, False))
    src = vs.Source(pull, peek, None)
    cfg = vs.default_config(**cfg_kw)
    h = ctypes.c_void_p()
    assert vs.lib.vstab_create(ctypes.byref(cfg), ctypes.byref(src), ctypes.byref(h)) == vs.OK, vs.lib.vstab_last_error()
    ow, oh = ctypes.c_int(), ctypes.c_int()
    assert vs.lib.vstab_get_output_info(h, ctypes.byref(ow), ctypes.byref(oh), None, None) == vs.OK
    outs = []
    while True:
        o = torch.empty((oh.value, ow.value, 3), dtype=torch.uint8, device=cuda)
        st = vs.lib.vstab_pull_frame(h, o.data_ptr(), o.stride(0))
        if st != vs.OK:
            break
        outs.append(o.cpu().numpy())
    msg = vs.lib.vstab_last_error()
    prof = vs.Profile()
    assert vs.lib.vstab_get_profile(h, ctypes.byref(prof)) == vs.OK
    vs.lib.vstab_destroy(h)
    return outs, st, msg, prof


def _open_fds():
    return len(os.listdir("/proc/self/fd"))


def test_dmabuf_frames_are_imported_and_read_in_place(vs, cuda, clip):
    """SURVEY.md 8(f) row 3 without libav: a decoder surface arrives as a DMA-BUF (what av_hwframe_map(..., DRM_PRIME) hands
    out: fd, size, per-plane offset and pitch).  A pool of device buffers is exported with
    hipMemGetHandleForAddressRange(DmaBufFd), every frame is handed over as {fd, size, offsets}; the library imports each
    object once and must produce the stream it produces from plain device pointers -- used in place (hold = forever) and
    copied (hold = 0) -- and one frame is compared with the checker as well (cvtColor -> reference createMap kernel -> remap).
    Replaces the VAAPI -> host -> OpenCL copies of AvFrameSourceMapOpenCl.cpp:17-66."""
    K, frames, _ = clip
    n = 14
    pool = _DmaBufPool(frames[:n])
    try:
        fds, size = pool.fds, pool.size
        ref_stab, ref = run_product(vs, cuda, frames[:n], smooth_radius=3, seed=9)
        Ko, (cw, ch) = oracle.get_output_camera(K, W, H)
        for hold in (1 << 29, 0):
            state = {"i": 0}

            def fill(o, advance):
                i = state["i"]
                if i >= n:
                    return vs.EOF
                o.mem, o.dmabuf_fd, o.dmabuf_size = 2, fds[i], size
                o.dmabuf_modifier = 0 if i % 2 else 0x00ffffffffffffff   # DRM_FORMAT_MOD_LINEAR / DRM_FORMAT_MOD_INVALID (implicit layout)
                o.y, o.uv = 64, 64 + W * H                   # byte offsets inside the object
                o.pitch_y = o.pitch_uv = W
                o.width, o.height, o.pts, o.hold, o.bit_depth = W, H, i, hold, 8
                if advance:
                    state["i"] += 1
                return 0
            outs, st, _, prof = _run_dmabuf_source(vs, cuda, fill, smooth_radius=3, seed=9)
            assert st == vs.EOF and len(outs) == len(ref) == n - 1
            assert prof.dmabuf_imports == n and prof.dmabuf_evictions == 0 and prof.dmabuf_cached == n
            for i, (a, b) in enumerate(zip(outs, ref)):
                assert np.array_equal(a, b), (hold, i)
            assert np.array_equal(outs[4], expect.warp(frames[5], oracle.map_params(K, Ko, ref_stab.warp_rotation(4)), cw, ch))
        # a frame whose planes do not fit in the object is refused, with a message

        def bad(o, advance):
            o.mem, o.dmabuf_fd, o.dmabuf_size = 2, fds[0], size
            o.y, o.uv, o.pitch_y, o.pitch_uv = 64, size - 16, W, W
            o.width, o.height, o.hold, o.bit_depth = W, H, 0, 8
            return 0
        outs, st, msg, _ = _run_dmabuf_source(vs, cuda, bad, smooth_radius=3, seed=9)
        assert not outs and st == vs.ERR_INVALID and b"DMA-BUF" in msg
    finally:
        pool.close()


def test_dmabuf_tiled_surfaces_are_refused_not_read_as_linear(vs, cuda, clip):
    """AVDRMObjectDescriptor.format_modifier travels in vstab_frame.dmabuf_modifier: anything but a linear layout (or "no
    modifier") is refused with VSTAB_ERR_UNSUPPORTED -- the same buffer that is accepted as linear would otherwise be read as
    rows of `pitch` bytes whatever its tiling (AvFrameSourceMapOpenCl.cpp:17-66 goes through hwframe transfers, which
    de-tile; an in-place import must not pretend)."""
    K, frames, _ = clip
    pool = _DmaBufPool(frames[:4])
    try:
        AMD_TILED = (0x02 << 56) | 0x1001   # a DRM_FORMAT_MOD_AMD value (vendor 0x02): some GFX9+ tiling
        for mod, ok in ((0, True), (AMD_TILED, False), (1 << 56 | 1, False)):
            state = {"i": 0}

            def fill(o, advance):
                i = state["i"]
                if i >= 4:
                    return vs.EOF
                o.mem, o.dmabuf_fd, o.dmabuf_size, o.dmabuf_modifier = 2, pool.fds[i], pool.size, mod
                o.y, o.uv, o.pitch_y, o.pitch_uv = 64, 64 + W * H, W, W
                o.width, o.height, o.pts, o.hold, o.bit_depth = W, H, i, 0, 8
                if advance:
                    state["i"] += 1
                return 0
            outs, st, msg, prof = _run_dmabuf_source(vs, cuda, fill, smooth_radius=1, tracking=0)
            if ok:
                assert st == vs.EOF and len(outs) == 3
            else:
                assert not outs and st == vs.ERR_UNSUPPORTED and b"modifier" in msg and b"LINEAR" in msg and prof.dmabuf_imports == 0
    finally:
        pool.close()


def test_dmabuf_import_leaves_no_descriptor_behind(vs, cuda, clip):
    """The import neither keeps nor needs the caller's descriptor (a duplicate per imported object used to leak).
    Upstream hands a FRESH dup() of the surface's fd with every frame and closes it as soon as the next callback comes -- so
    descriptor numbers are reused for different objects all the time (objects are recognised by their inode, not by the
    number) -- and the process holds exactly as many descriptors after the handle is destroyed as before it was created."""
    K, frames, _ = clip
    n = 14
    pool = _DmaBufPool(frames[:n])
    try:
        ref_stab, ref = run_product(vs, cuda, frames[:n], smooth_radius=3, seed=9)
        before = _open_fds()
        state = {"i": 0, "last": -1, "numbers": []}

        def fill(o, advance):
            if state["last"] >= 0:            # the descriptor handed out with the previous call: upstream is done with it
                os.close(state["last"])
                state["last"] = -1
            i = state["i"]
            if i >= n:
                return vs.EOF
            fd = os.dup(pool.fds[i])
            state["last"] = fd
            state["numbers"].append(fd)
            o.mem, o.dmabuf_fd, o.dmabuf_size, o.dmabuf_modifier = 2, fd, pool.size, 0
            o.y, o.uv, o.pitch_y, o.pitch_uv = 64, 64 + W * H, W, W
            o.width, o.height, o.pts, o.hold, o.bit_depth = W, H, i, 1 << 29, 8
            if advance:
                state["i"] += 1
            return 0
        outs, st, _, prof = _run_dmabuf_source(vs, cuda, fill, smooth_radius=3, seed=9)
        if state["last"] >= 0:
            os.close(state["last"])
        assert st == vs.EOF and len(outs) == n - 1 and prof.dmabuf_imports == n
        assert len(set(state["numbers"])) < n          # the same descriptor number did stand for different objects
        for i, (a, b) in enumerate(zip(outs, ref)):
            assert np.array_equal(a, b), i
        assert _open_fds() == before
    finally:
        pool.close()


def test_dmabuf_cache_eviction_and_reimport(vs, cuda, clip, monkeypatch):
    """More surfaces than the import cache holds (VSTAB_DMABUF_CACHE shrinks the 256-entry cache for the test): the least
    recently used object is unmapped -- never one a frame of the look-ahead window may still refer to -- and imported again
    when its turn comes round.  Every frame equals the plain-pointer stream, in place and copied."""
    K, frames, _ = clip
    n_pool, n = 30, 75
    seq = [frames[i % n_pool] for i in range(n)]
    pool = _DmaBufPool(frames[:n_pool])
    try:
        ref_stab, ref = run_product(vs, cuda, seq, smooth_radius=1, seed=9)
        monkeypatch.setenv("VSTAB_DMABUF_CACHE", "4")
        for hold in (1 << 29, 0):
            state = {"i": 0}

            def fill(o, advance):
                i = state["i"]
                if i >= n:
                    return vs.EOF
                o.mem, o.dmabuf_fd, o.dmabuf_size, o.dmabuf_modifier = 2, pool.fds[i % n_pool], pool.size, 0
                o.y, o.uv, o.pitch_y, o.pitch_uv = 64, 64 + W * H, W, W
                o.width, o.height, o.pts, o.hold, o.bit_depth = W, H, i, hold, 8
                if advance:
                    state["i"] += 1
                return 0
            outs, st, _, prof = _run_dmabuf_source(vs, cuda, fill, smooth_radius=1, seed=9)
            assert st == vs.EOF and len(outs) == n - 1
            # the ring has 18 slots: an object is kept for 20 pulls after its last use, so the cache settles at ~21 objects (not 4,
            # not 30) and every object is imported again on each pass over the pool
            assert prof.dmabuf_evictions >= n - n_pool and prof.dmabuf_imports == prof.dmabuf_evictions + prof.dmabuf_cached
            assert 4 < prof.dmabuf_cached < n_pool and prof.dmabuf_imports > n_pool
            for i, (a, b) in enumerate(zip(outs, ref)):
                assert np.array_equal(a, b), (hold, i)
    finally:
        pool.close()


def test_frame_lifetime_promise_hold(vs, cuda, clip):
    """vstab_frame.hold: a source that recycles one surface every pull (hold = 0, the default contract) and a source
    that keeps every frame alive (frames used in place, never copied) must produce the same stream."""
    K, frames, _ = clip
    n = 16
    ref_stab, ref = run_product(vs, cuda, frames[:n], smooth_radius=3, seed=9)      # ring source: frames used in place
    for hold, recycle in [(0, True), (2, False), (1 << 20, False)]:
        outs = _run_raw_device_source(vs, cuda, frames[:n], hold, recycle, smooth_radius=3, seed=9)
        assert len(outs) == len(ref) == n - 1, (hold, recycle)
        for i, (a, b) in enumerate(zip(outs, ref)):
            assert np.array_equal(a, b), (hold, recycle, i)
    # tracking off: nothing but the copy orders the stages
    ref_stab, ref = run_product(vs, cuda, frames[:n], smooth_radius=3, tracking=0)
    outs = _run_raw_device_source(vs, cuda, frames[:n], 0, True, smooth_radius=3, tracking=0)
    assert len(outs) == len(ref) and all(np.array_equal(a, b) for a, b in zip(outs, ref))


def test_finite_hold_borrowed_frames_wait_for_their_warp(vs, cuda, clip):
    """Frames used in place under a FINITE promise (a decoder pool of 18 surfaces, hold = 17 = r + read-ahead + 6 at
    r = 3): upstream counts pull callbacks, the warps run on the caller's stream.  With that stream kept ~20 ms behind
    the host per pull, the pool rewrites surface i % 18 while the warp of frame i - 18 is still queued unless the
    library waits for that warp before the callback that ends the promise."""
    K, frames, _ = clip
    n = 40
    ref_stab, ref = run_product(vs, cuda, frames[:n], smooth_radius=3, seed=9)
    for tracking in (1, 0):
        if not tracking:
            ref_stab, ref = run_product(vs, cuda, frames[:n], smooth_radius=3, tracking=0)
        outs = _run_raw_device_source(vs, cuda, frames[:n], 17, False, pool=18, lag=True, smooth_radius=3, seed=9, tracking=tracking)
        assert len(outs) == len(ref)
        for i, (a, b) in enumerate(zip(outs, ref)):
            assert np.array_equal(a, b), (tracking, i)


def test_upstream_error_surfaces_after_the_frames_read_ahead(vs, cuda, clip):
    """The reference meets an upstream error when it CONSUMES the failing frame (FrameSourceWarp.cpp:453-455): with
    frames 0..19 good and frame 20 failing at r = 3, outputs 0..15 are emitted (output k needs frames up to k + r + 1)
    and the pull of output 16 fails.  The read-ahead here calls upstream earlier; the error must not."""
    import ctypes
    import torch
    K, frames, _ = clip
    good = 20
    dev_frames = [torch.from_numpy(f).to(cuda) for f in frames[:good]]
    state = {"i": 0}

    def fill(out, advance):
        if state["i"] >= good:
            return 7
        f = dev_frames[state["i"]]
        o = out.contents
        o.y, o.uv = f.data_ptr(), f.data_ptr() + H * f.stride(0)
        o.pitch_y = o.pitch_uv = f.stride(0)
        o.width, o.height, o.mem, o.pts, o.hold = W, H, 0, 0, 1 << 30
        if advance:
            state["i"] += 1
        return 0
    pull = vs.PULL_FN(lambda u, o: fill(o, True))
    peek = vs.PULL_FN(lambda u, o: fill(o, False))
    src = vs.Source(pull, peek, None)
    cfg = vs.default_config(smooth_radius=3, seed=9)
    h = ctypes.c_void_p()
    assert vs.lib.vstab_create(ctypes.byref(cfg), ctypes.byref(src), ctypes.byref(h)) == vs.OK
    ow, oh = ctypes.c_int(), ctypes.c_int()
    assert vs.lib.vstab_get_output_info(h, ctypes.byref(ow), ctypes.byref(oh), None, None) == vs.OK
    ref_stab, ref = run_product(vs, cuda, frames[:good + 8], smooth_radius=3, seed=9)
    emitted = 0
    while True:
        o = torch.empty((oh.value, ow.value, 3), dtype=torch.uint8, device=cuda)
        st = vs.lib.vstab_pull_frame(h, o.data_ptr(), o.stride(0))
        if st != vs.OK:
            break
        assert np.array_equal(o.cpu().numpy(), ref[emitted]), emitted
        emitted += 1
    assert st == vs.ERR_SOURCE and b"7" in vs.lib.vstab_last_error()
    assert emitted == good - 3 - 1
    o = torch.empty((oh.value, ow.value, 3), dtype=torch.uint8, device=cuda)
    assert vs.lib.vstab_pull_frame(h, o.data_ptr(), o.stride(0)) == vs.ERR_SOURCE      # sticky
    vs.lib.vstab_destroy(h)


def test_cpp_adapter_example_runs(vs, cuda):
    """examples/display_image.cpp: the reference's DisplayImage loop on include/vstab_frame_source.hpp (same class
    name, constructor arguments and thrown-int EOF).  Built by __graft_entry__.build(); exit code 0 = n - 1 frames."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "display_image")
    if not os.path.exists(exe):
        pytest.skip("examples/display_image not built (run __graft_entry__.build())")
    r = subprocess.run([exe, "50"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "49 frames of" in r.stderr and "opencv-warped:" in r.stderr          # Profiler.cpp:25-34 output format
    # the gyro path the reference stubs (gpmf.cpp:5-11): samples -> vstab_gyro_integrate -> delta / read-out rotations per frame
    r = subprocess.run([exe, "40", "gyro"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "39 frames of" in r.stderr, r.stderr
    # frames handed on as NV12 planes, remapped plane-wise (FrameSourceWarp::pull_frame_nv12(..., plane_wise = true)): the encoder's input
    r = subprocess.run([exe, "40", "nv12"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "39 frames of" in r.stderr, r.stderr


def test_p010_input_equals_8bit_input_of_the_truncated_frames(vs, cuda, clip):
    """BASELINE config 5 input format: 16-bit planes are narrowed (sample >> 8) on ingest, then the 8-bit path runs."""
    K, frames, _ = clip
    n = 12
    ref_stab, ref = run_product(vs, cuda, frames[:n], smooth_radius=3, seed=9)
    outs = _run_raw_device_source(vs, cuda, frames[:n], 1 << 20, False, p010=True, smooth_radius=3, seed=9)
    assert len(outs) == len(ref) and all(np.array_equal(a, b) for a, b in zip(outs, ref))
    # a decoder that recycles ONE P010 surface (hold = 0): the library may call upstream again as soon as the narrowing copy of the
    # surface is through (an event behind the copies, not the frame's pyramid) -- and not before
    outs = _run_raw_device_source(vs, cuda, frames[:n], 0, True, p010=True, smooth_radius=3, seed=9)
    assert len(outs) == len(ref) and all(np.array_equal(a, b) for a, b in zip(outs, ref))


def test_two_handles_interleaved_are_independent(vs, cuda, clip):
    """One handle = one clip; handles share nothing: pulling two clips alternately gives
    each the stream it gives alone."""
    import torch
    K, frames, _ = clip
    a_frames, b_frames = frames[:14], [np.ascontiguousarray(f[:, ::-1]) for f in frames[3:17]]   # second clip: mirrored, shifted in time
    _, ref_a = run_product(vs, cuda, a_frames, smooth_radius=3, seed=1)
    _, ref_b = run_product(vs, cuda, b_frames, smooth_radius=4, seed=2)
    sa = vs.Stabilizer([torch.from_numpy(f).to(cuda) for f in a_frames], total=len(a_frames), smooth_radius=3, seed=1)
    sb = vs.Stabilizer([torch.from_numpy(f).to(cuda) for f in b_frames], total=len(b_frames), smooth_radius=4, seed=2)
    outs_a, outs_b = [], []
    done_a = done_b = False
    while not (done_a and done_b):
        if not done_a:
            o = sa.pull()
            done_a = o is None
            if o is not None:
                outs_a.append(o.cpu().numpy())
        if not done_b:
            o = sb.pull()
            done_b = o is None
            if o is not None:
                outs_b.append(o.cpu().numpy())
    assert len(outs_a) == len(ref_a) and all(np.array_equal(x, y) for x, y in zip(outs_a, ref_a))
    assert len(outs_b) == len(ref_b) and all(np.array_equal(x, y) for x, y in zip(outs_b, ref_b))


def test_external_rotation_source_replaces_optical_flow(vs, cuda, clip):
    """SURVEY.md 8(f) row 4 (the gyro path the reference stubs): with tracking off, upstream's per-frame rotation is
    accumulated (:441), smoothed (:444, :471) and corrected (:472-475) exactly as an optical-flow estimate would be."""
    import ctypes
    import torch
    K, frames, rots = clip
    n, r = 14, 3
    dev = [torch.from_numpy(f).to(cuda) for f in frames[:n]]
    deltas = [np.eye(3)] + [rots[k] @ rots[k - 1].T for k in range(1, n)]     # what a gyro integrates between frames
    keep = [np.ascontiguousarray(d, np.float64) for d in deltas]
    state = {"i": 0}

    def fill(out, advance):
        i = state["i"]
        if i >= n:
            return vs.EOF
        t = dev[i]
        o = out.contents
        o.y, o.uv = t.data_ptr(), t.data_ptr() + H * t.stride(0)
        o.pitch_y = o.pitch_uv = t.stride(0)
        o.width, o.height, o.mem, o.pts, o.hold, o.bit_depth = W, H, 0, i, 1 << 20, 8
        o.delta_rotation = keep[i].ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        if advance:
            state["i"] += 1
        return 0
    pull = vs.PULL_FN(lambda u, o: fill(o, True))
    peek = vs.PULL_FN(lambda u, o: fill(o, False))
    src = vs.Source(pull, peek, None)
    cfg = vs.default_config(smooth_radius=r, tracking=0)
    h = ctypes.c_void_p()
    assert vs.lib.vstab_create(ctypes.byref(cfg), ctypes.byref(src), ctypes.byref(h)) == vs.OK
    Ko, (cw, ch) = oracle.get_output_camera(K, W, H)
    outs, warp_R = [], []
    while True:
        o = torch.empty((ch, cw, 3), dtype=torch.uint8, device=cuda)
        st = vs.lib.vstab_pull_frame(h, o.data_ptr(), o.stride(0))
        if st == vs.EOF:
            break
        assert st == vs.OK, vs.lib.vstab_last_error()
        outs.append(o.cpu().numpy())
        R = np.zeros(9)
        assert vs.lib.vstab_get_warp_rotation(h, len(outs) - 1, R.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == vs.OK
        warp_R.append(R.reshape(3, 3))
    vs.lib.vstab_destroy(h)
    assert len(outs) == n - 1
    # the oracle's state machine fed the same rotations
    it = iter(deltas[1:])
    rec = []
    sm = oracle.WarpStateMachine(frames[:n], r, lambda g: np.zeros((200, 2), np.float32), lambda p, c, pts: (pts, pts),
                                 lambda pp, cp: (next(it), 100), lambda f, R: (rec.append(R), f)[1])
    exp_frames = []
    while True:
        o = sm.pull_frame()
        if o is None:
            break
        exp_frames.append(o)
    assert len(rec) == n - 1
    for i in range(n - 1):
        assert np.allclose(warp_R[i], rec[i], atol=1e-11), i
    for i in (0, 5, n - 2):
        p = oracle.map_params(K, Ko, warp_R[i])
        assert np.array_equal(outs[i], expect.warp(exp_frames[i], p, cw, ch)), i
    # the correction really follows the sensor: it is not the identity
    assert max(oracle.rotation_angle(R) for R in warp_R) > 1e-3


@pytest.mark.parametrize("prec", [expect.OPENCL, expect.IEEE])
def test_rolling_shutter_readout_rotation_reaches_the_warp(vs, cuda, clip, prec):
    """BASELINE config 5 in the pipeline object: a sensor source that also reports the camera's rotation during the frame's
    read-out.  The stabilising rotation W is what it would be without it (the smoother sees the per-frame deltas only); the
    frame is warped with W for its first row and readout * W for its last (oracle.warp_nv12_rs, the definition of
    vstab_warp_nv12_rs).  Frames without a read-out rotation take the ordinary warp.  No reference counterpart: the
    reference's gyro path is a stub (gpmf.cpp:5-11)."""
    import ctypes
    import torch
    K, frames, rots = clip
    n, r = 12, 3
    dev = [torch.from_numpy(f).to(cuda) for f in frames[:n]]
    deltas = [np.eye(3)] + [rots[k] @ rots[k - 1].T for k in range(1, n)]
    readouts = [oracle.rodrigues(np.array([0.004 * np.sin(k), -0.003 * np.cos(2 * k), 0.006 * np.sin(0.5 * k + 1)])) for k in range(n)]
    keep_d = [np.ascontiguousarray(d, np.float64) for d in deltas]
    keep_r = [np.ascontiguousarray(d, np.float64) for d in readouts]
    dp = ctypes.POINTER(ctypes.c_double)

    def run(with_readout):
        state = {"i": 0}

        def fill(out, advance):
            i = state["i"]
            if i >= n:
                return vs.EOF
            t = dev[i]
            o = out.contents
            o.y, o.uv = t.data_ptr(), t.data_ptr() + H * t.stride(0)
            o.pitch_y = o.pitch_uv = t.stride(0)
            o.width, o.height, o.mem, o.pts, o.hold, o.bit_depth = W, H, 0, i, 1 << 30, 8
            o.delta_rotation = keep_d[i].ctypes.data_as(dp)
            o.readout_rotation = keep_r[i].ctypes.data_as(dp) if with_readout(i) else None
            if advance:
                state["i"] += 1
            return 0
        pull = vs.PULL_FN(lambda u, o: fill(o, True))
        peek = vs.PULL_FN(lambda u, o: fill(o, False))
        src = vs.Source(pull, peek, None)
        cfg = vs.default_config(smooth_radius=r, tracking=0, map_precision=prec)
        h = ctypes.c_void_p()
        assert vs.lib.vstab_create(ctypes.byref(cfg), ctypes.byref(src), ctypes.byref(h)) == vs.OK
        outs, warp_R = [], []
        while True:
            o = torch.empty((ch, cw, 3), dtype=torch.uint8, device=cuda)
            st = vs.lib.vstab_pull_frame(h, o.data_ptr(), o.stride(0))
            if st == vs.EOF:
                break
            assert st == vs.OK, vs.lib.vstab_last_error()
            outs.append(o.cpu().numpy())
            R = np.zeros(9)
            assert vs.lib.vstab_get_warp_rotation(h, len(outs) - 1, R.ctypes.data_as(dp)) == vs.OK
            warp_R.append(R.reshape(3, 3))
        vs.lib.vstab_destroy(h)
        return outs, warp_R
    Ko, (cw, ch) = oracle.get_output_camera(K, W, H)
    plain, plain_R = run(lambda i: False)
    outs, warp_R = run(lambda i: i % 4 != 2)            # frames 2, 6, 10: global shutter
    assert len(outs) == len(plain) == n - 1
    differ = 0
    for i in range(n - 1):
        assert np.allclose(warp_R[i], plain_R[i], atol=1e-12), i   # the read-out rotation does not enter the smoother
        k = i + 1                                                  # output i is input frame i + 1
        p = oracle.map_params(K, Ko, warp_R[i])
        if k % 4 != 2:
            pb = oracle.map_params(K, Ko, readouts[k] @ warp_R[i])
            assert np.array_equal(outs[i], expect.warp(frames[k], p, cw, ch, prec, rot_bottom=pb[8:])), i
            differ += not np.array_equal(outs[i], plain[i])
        else:
            assert np.array_equal(outs[i], plain[i]), i
            assert np.array_equal(outs[i], expect.warp(frames[k], p, cw, ch, prec)), i
    assert differ >= 6                                             # it is not a no-op


def _check_against_oracle_state_machine(vs, cuda, frames, K, w, h, r, seed):
    """Whole-pipeline equivalence on an arbitrary clip: decisions, counts, rotations handed to the warp, pixels."""
    stab, outs = run_product(vs, cuda, frames, smooth_radius=r, seed=seed)
    log = stab.frame_log()
    n = len(frames)
    assert len(outs) == max(n - 1, 0) and len(log) == max(n - 1, 0)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    rng = oracle.Pcg32(seed)  # the clip's stream of vstab_config.seed
    counts, warp_rots = [], []

    def track(prev, cur, corners):
        if len(corners) == 0:
            counts.append((0, 0))
            return corners, corners
        nxt, st = oracle.pyr_lk(prev, cur, corners)
        counts.append((len(corners), int((st > 0).sum())))
        return corners[st > 0], nxt[st > 0]

    sm = oracle.WarpStateMachine(frames, r, lambda g: oracle.good_features(np.ascontiguousarray(g)), track,
                                 lambda pp, cp: oracle.estimate_rotation(pp, cp, K, Ko, rng), lambda f, R: (warp_rots.append(R), f)[1])
    exp = []
    while True:
        o = sm.pull_frame()
        if o is None:
            break
        exp.append(o)
    assert [l["key"] for l in log] == [l["key"] for l in sm.log]
    assert [(l["n_corners"], l["n_tracked"]) for l in log] == counts
    assert [l["inliers"] for l in log] == [l["inliers"] for l in sm.log]   # the oracle's own RANSAC, same seeded stream
    assert all(np.allclose(a["R"], b["R"], atol=1e-9) for a, b in zip(log, sm.log))
    for i in range(len(outs)):
        assert np.allclose(stab.warp_rotation(i), warp_rots[i], atol=1e-9), i
        p = oracle.map_params(K, Ko, stab.warp_rotation(i))
        assert np.array_equal(outs[i], expect.warp(exp[i], p, cw, ch)), i
    return log


def test_pipeline_degenerate_clips(vs, cuda):
    """Featureless frames (no corners at all: every frame is a key frame, every estimate falls back), a clip with a
    handful of corners, sizes that are even but not multiples of four, and a two-frame clip."""
    w, h = 322, 182
    K = oracle.get_preset_camera(4, w, h)
    black = [np.full((h * 3 // 2, w), 128, np.uint8) for _ in range(7)]
    for f in black:
        f[:h] = 16
    log = _check_against_oracle_state_machine(vs, cuda, black, K, w, h, 2, 1)
    assert all(l["key"] and l["n_corners"] == 0 and l["fallback"] for l in log)
    few = []
    for k in range(8):
        f = np.full((h * 3 // 2, w), 128, np.uint8)
        f[:h] = 40
        for (x, y) in [(60, 40), (200, 90), (120, 140)]:
            f[y + k:y + k + 25, x + 2 * k:x + 2 * k + 30] = 220          # three moving rectangles: a dozen corners
        few.append(f)
    log = _check_against_oracle_state_machine(vs, cuda, few, K, w, h, 2, 2)
    assert all(0 < l["n_corners"] < 40 and l["fallback"] for l in log)
    frames, _ = synth.shaky_clip(5, K, w, h, 9, sigma=0.004)
    _check_against_oracle_state_machine(vs, cuda, frames, K, w, h, 3, 3)
    _check_against_oracle_state_machine(vs, cuda, frames[:2], K, w, h, 3, 4)
    _check_against_oracle_state_machine(vs, cuda, frames[:1], K, w, h, 3, 5)


@pytest.mark.parametrize("env", [{}, {"VSTAB_LK_SEGMENT": "1"}, {"VSTAB_LK_SEGMENT": "3", "VSTAB_PREFETCH": "2"}, {"VSTAB_PREFETCH": "16", "VSTAB_LK_SEG_TARGET": "8"},
                                 {"VSTAB_CHAIN_LK": "0", "VSTAB_PREFETCH": "1"}, {"VSTAB_EPOCH_OVERLAP": "0"}, {"VSTAB_EPOCH_OVERLAP": "1", "VSTAB_PREFETCH": "16"}])
def test_every_clip_length_and_radius_against_the_state_machine(vs, cuda, monkeypatch, env):
    """Clips of 1 .. 26 frames (shorter than, equal to and longer than the look-ahead; across the 21-frame key-frame counter) with
    smoothing radii 1, 2, 5 (radius 0 has no reference behaviour: gram_sg divides by zero there), under the launch-ahead settings a handle can be created with (frames per tracker launch, read-ahead depth,
    no launches ahead at all): the end-of-stream padding, the key-frame rule and the tracker segments meet in every combination, and
    every decision, count, rotation and pixel has to be the oracle state machine's."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w, h = 322, 182
    K = oracle.get_preset_camera(4, w, h)
    frames, _ = synth.shaky_clip(17, K, w, h, 26, sigma=0.004)
    lengths = [1, 2, 3, 4, 5, 6, 9, 21, 22, 23, 26] if not env else [1, 2, 5, 9, 22, 26]
    for r in (1, 2, 5):
        for n in lengths:
            _check_against_oracle_state_machine(vs, cuda, frames[:n], K, w, h, r, 100 + n)


@pytest.mark.parametrize("env", [None, "1", "0"])
def test_caller_on_a_stream_of_its_own(vs, cuda, clip, monkeypatch, env):
    """vstab_config.stream: a caller that works on a stream of its own gets the frames of the default-stream run, bit for bit -- with the
    speculative corner detection on the read-ahead stream (the default beside a non-NULL stream: INTEGRATION.md section 3), on a stream
    of its own (VSTAB_DETECT_STREAM=1) and without one on the default stream (=0).  26 frames: one planned key frame inside."""
    import torch
    K, frames, _ = clip
    _, ref = run_product(vs, cuda, frames[:26], smooth_radius=3, seed=9)
    if env is not None:
        monkeypatch.setenv("VSTAB_DETECT_STREAM", env)
    if env == "0":
        _, outs = run_product(vs, cuda, frames[:26], smooth_radius=3, seed=9)
    else:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            _, outs = run_product(vs, cuda, frames[:26], smooth_radius=3, seed=9)
        s.synchronize()
    assert len(outs) == len(ref) == 25 and all(np.array_equal(a, b) for a, b in zip(outs, ref))


def test_corner_selection_by_the_caller_when_the_helper_thread_wakes_up_late(vs, cuda, monkeypatch):
    """The speculative detection's corners are selected by a helper thread that sleeps between two detections; when it has not woken up by
    the time the corners are looked for, the caller selects them itself.  720p, 70 frames of the bench's clip: 200 corners survive, so the
    key frames are the ones the counter plans (every 21st frame) and their detections run ahead, speculatively.  A helper that wakes up
    300 ms late (VSTAB_SPEC_HELPER_DELAY_US) leaves every one of those selections to the caller: the frames and the key-frame list are
    those of the run in which the helper selects, and the handle's counters say who selected (no wall-clock bound)."""
    import torch
    import bench
    w, h, n = 1280, 720, 70
    K = oracle.get_preset_camera(4, w, h)
    dev_frames, _ = bench.shaky_ring(torch, cuda, w, h, K, n, seed=5)

    def run():
        stab = vs.Stabilizer(dev_frames, total=n, smooth_radius=2, seed=4)
        outs = []
        while True:
            o = stab.pull()
            if o is None:
                break
            outs.append(o.cpu().numpy())
        return stab, outs
    ref_stab, ref = run()
    keys = [k for k, l in enumerate(ref_stab.frame_log()) if l["key"]]
    assert len(keys) >= 3 and len({b - a for a, b in zip(keys, keys[1:])}) == 1, keys   # the counter's key frames only: planned, hence speculated
    base = ref_stab.profile()
    assert base["corner_selections_by_caller"] + base["corner_selections_by_helper"] >= len(keys), base
    monkeypatch.setenv("VSTAB_SPEC_HELPER_DELAY_US", "300000")   # (read when the handle's tracker is constructed)
    stab, outs = run()
    assert len(outs) == len(ref) == n - 1 and all(np.array_equal(a, b) for a, b in zip(outs, ref))
    assert [l["key"] for l in stab.frame_log()] == [l["key"] for l in ref_stab.frame_log()]
    late = stab.profile()
    assert late["corner_selections_by_caller"] >= len(keys) and late["corner_selections_by_helper"] == 0, late


def test_epochs_in_turn_second_stream_is_used_and_changes_nothing(vs, cuda, monkeypatch):
    """What follows a planned key frame (its speculative detection, the tracker launch from those corners, the launches chained behind it) depends
    on nothing tracked before it; for frames up to 1920 x 1200 it runs on the second of the handle's two epoch streams beside the epoch still
    being tracked (the tracker's chain sets the frame period there).  720p, 110 frames of the bench's clip: the handle's counter says the second
    stream took epochs, and frames, key frames and per-frame counts are those of the run with everything on one stream (VSTAB_EPOCH_OVERLAP=0)
    -- and of a run with the deepest read-ahead, where the two chains overlap longest."""
    import torch
    import bench
    w, h, n = 1280, 720, 110
    K = oracle.get_preset_camera(4, w, h)
    dev_frames, _ = bench.shaky_ring(torch, cuda, w, h, K, n, seed=7)

    def run():
        stab = vs.Stabilizer(dev_frames, total=n, smooth_radius=2, seed=4)
        outs = []
        while True:
            o = stab.pull()
            if o is None:
                break
            outs.append(o.cpu().numpy())
        return stab, outs
    stab, outs = run()
    log = [(l["key"], l["n_corners"], l["n_tracked"], l["inliers"]) for l in stab.frame_log()]
    keys = [k for k, l in enumerate(log) if l[0]]
    assert len(keys) >= 5, keys
    assert stab.profile()["epochs_in_turn"] >= 2, stab.profile()          # every other planned key frame, once launches run ahead
    for env, expect_second in (({"VSTAB_EPOCH_OVERLAP": "0"}, False), ({"VSTAB_PREFETCH": "16"}, True)):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        stab2, outs2 = run()
        for k in env:
            monkeypatch.delenv(k)
        assert (stab2.profile()["epochs_in_turn"] > 0) == expect_second, (env, stab2.profile())
        assert [(l["key"], l["n_corners"], l["n_tracked"], l["inliers"]) for l in stab2.frame_log()] == log, env
        assert len(outs2) == len(outs) == n - 1 and all(np.array_equal(a, b) for a, b in zip(outs2, outs)), env


def test_pull_into_host_memory_equals_device_pull(vs, cuda, clip):
    import torch
    K, frames, _ = clip
    _, ref = run_product(vs, cuda, frames[:9], smooth_radius=2, seed=6)
    stab = vs.Stabilizer([torch.from_numpy(f).to(cuda) for f in frames[:9]], total=9, smooth_radius=2, seed=6)
    outs = []
    while True:
        o = stab.pull_host()
        if o is None:
            break
        outs.append(o)
    assert len(outs) == len(ref) and all(np.array_equal(a, b) for a, b in zip(outs, ref))


def test_kalman_smoother_mode_matches_its_definition(vs, cuda, clip):
    """the Kalman mode (constants of init_filter / kalman.cpp, never called by the reference) --
    product vs the oracle's matrix-form cv::KalmanFilter on the product's own accumulated rotations."""
    K, frames, _ = clip
    stab, outs = run_product(vs, cuda, frames[:16], smooth_radius=2, smoother=vs.SMOOTHER_KALMAN, seed=3)
    log = stab.frame_log()
    kf = oracle.KalmanRotationFilter()
    for i, lg in enumerate(log):
        corrected = kf.update(lg["R_accum"])
        expect = np.linalg.inv(corrected @ np.linalg.inv(lg["R_accum"]))      # :472, :475
        assert np.allclose(stab.warp_rotation(i), expect, atol=1e-10), i
    assert oracle.rotation_angle(stab.warp_rotation(10)) > 1e-4               # it does smooth: the correction is not the identity


def test_pipeline_at_4k_baseline_config(vs, cuda):
    """BASELINE config 3 geometry (3840x2160, WIDE169_MEASURED -> 3524x1999, the headline workload) end to end against
    the oracle's state machine: key-frame decisions, corner and track counts, warp rotations, every emitted frame
    bit-exact; rotations close to ground truth; NV12 pull of the same clip = converted BGR."""
    w, h, r, n = 3840, 2160, 1, 4
    K = oracle.get_preset_camera(4, w, h)
    frames, rots = synth.shaky_clip(11, K, w, h, n, sigma=0.002)
    log = _check_against_oracle_state_machine(vs, cuda, frames, K, w, h, r, 2)
    for k, lg in enumerate(log, start=1):
        assert lg["inliers"] >= 40 and oracle.rotation_angle(lg["R"] @ (rots[k] @ rots[k - 1].T).T) < 3e-3
    import torch
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    assert (cw, ch) == (3524, 1999)
    s2 = vs.Stabilizer([torch.from_numpy(f).to(cuda) for f in frames], total=n, smooth_radius=r, seed=2)
    y, uv = s2.pull_nv12()
    bgr = expect.warp(frames[1], oracle.map_params(K, Ko, s2.warp_rotation(0)), cw, ch)
    ey, euv = oracle.cvt_bgr_nv12(bgr)
    assert np.array_equal(y.cpu().numpy(), ey) and np.array_equal(uv.cpu().numpy().reshape(euv.shape), euv)


def test_long_bench_shaped_run_1080p(vs, cuda):
    """The configuration bench.py times, checked end to end: 1080p, smooth_radius 30, 160 frames through the C ring source
    (frames used in place: `hold`), chained tracker launches and speculative corner detection on -- against the oracle's
    state machine with the oracle's own detector, tracker and rotation estimator: key-frame list (the > 20 frames rule
    fires seven times), corner / track / inlier counts, rotations, every tenth emitted frame bit for bit."""
    _bench_shaped_run(vs, cuda, 1920, 1080, 30, 160, 7)


def test_record_handoff_stress_three_runs_of_6000_frames_agree(vs, cuda):
    """The tracker's results reach the host (and the next chained launch) as self-tagged 8-byte granules with no fence
    (argued, never stress-verified).  Stress: the bench's 1080p pipeline over 6000
    frames -- 6000 chained launches x 200 records, speculative detections, helper threads -- three times over the same
    clip.  A torn or stale record anywhere would move a point, hence a rotation; a decision taken on a different schedule
    would move a key frame.  Every run must report the same key frames, the same track / inlier counts and bit-identical
    rotations for every frame, and the same output bytes for every 97th frame."""
    import torch
    import zlib
    import bench
    w, h, n = 1920, 1080, 6000
    K = oracle.get_preset_camera(4, w, h)
    dev_frames, _ = bench.shaky_ring(torch, cuda, w, h, K, 64, seed=9)
    runs = []
    for _ in range(3):
        stab = vs.Stabilizer(dev_frames, total=n, smooth_radius=30, seed=321)
        crcs, rots, i = [], [], 0
        while True:
            o = stab.pull()
            if o is None:
                break
            if i % 97 == 0:
                crcs.append(zlib.crc32(o.cpu().numpy().tobytes()))
            rots.append(stab.warp_rotation(i).copy())
            i += 1
        assert i == n - 1
        log = stab.frame_log()
        runs.append((crcs, np.stack(rots), [(e["key"], e["n_corners"], e["n_tracked"], e["inliers"]) for e in log]))
        stab.close()
    keys = sum(1 for e in runs[0][2] if e[0])
    assert keys >= 270, keys   # the > 20 frames rule fires about every 20th pair
    for r in runs[1:]:
        assert r[2] == runs[0][2]
        assert np.array_equal(r[1], runs[0][1])
        assert r[0] == runs[0][0]


def test_bench_shaped_run_4k_config_3(vs, cuda):
    """BASELINE config 3 at its own size through the whole pipeline object: 4K, smooth_radius 30, 52 frames of the bench's
    clip, frames used in place, chained tracker launches through two counter-triggered key frames and their speculative
    detections -- every decision, count and rotation against the oracle's state machine, every tenth frame bit for bit."""
    _bench_shaped_run(vs, cuda, 3840, 2160, 30, 52, 2)


def _bench_shaped_run(vs, cuda, w, h, r, n, min_keys, transform=None):
    import torch
    import bench
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    dev_frames, _ = bench.shaky_ring(torch, cuda, w, h, K, n, seed=5)
    if transform:
        dev_frames = transform(dev_frames)
        n = len(dev_frames)
    stab = vs.Stabilizer(dev_frames, total=n, smooth_radius=r, seed=77)
    outs = {}
    i = 0
    while True:
        o = stab.pull()
        if o is None:
            break
        if i % 10 == 0 or i == n - 2:
            outs[i] = o.cpu().numpy()
        i += 1
    assert i == n - 1
    log = stab.frame_log()
    frames = [f.cpu().numpy() for f in dev_frames]
    rng = oracle.Pcg32(77)
    counts, warp_rots = [], []

    def track(prev, cur, corners):
        nxt, st = oracle.pyr_lk(prev, cur, corners)
        counts.append((len(corners), int((st > 0).sum())))
        return corners[st > 0], nxt[st > 0]

    sm = oracle.WarpStateMachine(frames, r, lambda g: oracle.good_features(np.ascontiguousarray(g)), track,
                                 lambda pp, cp: oracle.estimate_rotation(pp, cp, K, Ko, rng), lambda f, R: (warp_rots.append(R), f)[1])
    exp = []
    while True:
        o = sm.pull_frame()
        if o is None:
            break
        exp.append(o)
    assert len(exp) == n - 1
    keys = [k for k, l in enumerate(log) if l["key"]]
    assert keys == [k for k, l in enumerate(sm.log) if l["key"]] and len(keys) >= min_keys
    assert [(l["n_corners"], l["n_tracked"]) for l in log] == counts
    assert [l["inliers"] for l in log] == [l["inliers"] for l in sm.log]
    assert all(np.allclose(a["R"], b["R"], atol=1e-9) for a, b in zip(log, sm.log))
    for k in range(n - 1):
        assert np.allclose(stab.warp_rotation(k), warp_rots[k], atol=1e-9), k
    for k, got in outs.items():
        assert np.array_equal(got, expect.warp(exp[k], oracle.map_params(K, Ko, stab.warp_rotation(k)), cw, ch)), k
    return log


def test_unplanned_key_frames_drop_the_tracker_launches_enqueued_ahead(vs, cuda, monkeypatch):
    """The tracker runs ahead of the host in launches that cover several frames each, assuming that the only key frames are the
    ones the counter predicts (:415, every 21st frame).  Here the clip fades to a low-contrast version of itself for a few
    frames (1080p, 200 corners): half of the features fail the tracker's eigenvalue test there, so the COUNT half of the rule
    (< 150 survivors) fires at frames nobody planned for -- what was enqueued ahead must be dropped, corners re-detected and
    tracking restarted from them, with every decision, count, rotation and pixel equal to the oracle's state machine, which
    knows nothing of launches.  Then the same clip with one frame per launch, with short segments, and with no launches
    ahead at all: the same log."""
    w, h = 1920, 1080

    def fade(frames):
        out = list(frames[:25])
        for f in frames[25:31]:
            g = f.clone()
            g[:h] = (96 + torch_floor_div(g[:h].to(torch_int32()) - 96, 12)).to(g.dtype)
            out.append(g)
        return out + list(frames[31:52])

    import torch

    def torch_int32():
        return torch.int32

    def torch_floor_div(a, b):
        return torch.div(a, b, rounding_mode="floor")
    log = _bench_shaped_run(vs, cuda, w, h, 5, 64, 3, transform=fade)
    keys = [i for i, l in enumerate(log) if l["key"]]
    assert keys[0] == 20                                        # log index i is frame i + 1: the counter's key frame (frame 21)
    assert sum(24 <= k <= 34 for k in keys) >= 2, keys          # and key frames nobody planned for: into the fade and out of it
    ref = [(l["key"], l["n_corners"], l["n_tracked"], l["inliers"]) for l in log]
    # (1080p: planned key frames' detection and tracking alternate between two streams by default -- epochs in turn; "0": one stream)
    for env in ({"VSTAB_LK_SEGMENT": "1"}, {"VSTAB_CHAIN_LK": "0"}, {"VSTAB_LK_SEGMENT": "3", "VSTAB_LK_SEG_TARGET": "2"}, {"VSTAB_EPOCH_OVERLAP": "0"},
                {"VSTAB_EPOCH_OVERLAP": "1", "VSTAB_PREFETCH": "16"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        log2 = _bench_shaped_run(vs, cuda, w, h, 5, 64, 3, transform=fade)
        for k in env:
            monkeypatch.delenv(k)
        assert [(l["key"], l["n_corners"], l["n_tracked"], l["inliers"]) for l in log2] == ref, env


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`bench.py --gpus 2` outside torch.distributed.run starts two ranks itself (rehearsed on one GPU over gloo) and
    rank 0 reports n_gpus = 2, the parity check of an emitted frame and the pre-roll."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--workload", "1080p",
                        "--steps", "20", "--warmup", "5", "--batch", "8", "--preroll", "40", "--fixed-preroll", "--ring", "16", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["preroll"] == 40 and line["parity_check"] == "ok" and line["scaling"] == "weak"
    assert line["roofline"]["kernel"] == "k_warp_fused" and line["value"] > 0


def test_bench_four_ranks_share_one_gpu():
    """The multi-rank path with as many ranks as one card admits: the GPU box's process guard allows six processes with the GPU
    open, and this test process and torch.distributed.run's agent are two of them -- `bench.py --gpus 4 --share-gpu
    --dist-backend gloo`: four pipelines side by side on one GPU, rendezvous, pinning to disjoint CPU sets, the gather of four
    records, the 4-entry concat list, parity of rank 0's frame.  Functional only: four clips on one card say nothing about
    scaling (N = 8 is rehearsed without GPUs in tests/test_shard_cpu.py; the real curve is the driver's)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    shape = ["--workload", "1080p", "--steps", "4", "--warmup", "1", "--batch", "64", "--preroll", "200", "--fixed-preroll", "--ring", "16", "--no-cpu-baseline",
             "--skip-copy-pass"]

    def bench_line(extra):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra + shape, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        return json.loads(lines[0])
    one = bench_line(["--gpus", "1"])
    line = bench_line(["--gpus", "4", "--share-gpu", "--dist-backend", "gloo"])
    assert line["n_gpus"] == 4 and line["config"]["clips"] == 4 and line["parity_check"] == "ok" and len(line["rank_cpus_all"]) == 4
    # what a rank costs the host: the CPUs it keeps busy (its frame loop and the handle's two helper threads spin before they sleep).
    # Four ranks on ONE card run at a quarter of the rate each, so their CPU-seconds PER FRAME are four times the single rank's; what
    # has to hold for N ranks to fit a host is that a rank's busy CPUs do not grow with N and stay inside the CPUs it is pinned to.
    busy1, busy4 = one["host"]["cpus_busy"], line["host"]["all_ranks_cpus_busy"]
    assert one["host"]["cpu_seconds_per_1000_frames"] > 0 and one["host"]["threads"] >= 4 and one["host"]["pinned_cpus"] == 8
    # (bound: 1.5 x the single rank's figure, or the four threads of a rank that work at all -- frame loop, estimate worker, corner-selection
    #  helper, runtime -- whichever is larger: a single rank that waits less spins less, 1.5 - 2.8 busy CPUs from run to run)
    assert len(busy4) == 4 and all(0 < b <= max(1.5 * busy1, 4.0) and b <= 8 for b in busy4), (busy1, busy4)
    sets = []
    for txt in line["rank_cpus_all"]:
        first, last = (int(v) for v in txt.split(" ")[0].split("-"))
        sets.append(set(range(first, last + 1)))
    assert all(not (a & b) for i, a in enumerate(sets) for b in sets[i + 1:]), line["rank_cpus_all"]
    assert open(os.path.join(root, "gpurun_out", "concat_list_4gpu.txt")).read().count("file '") == 4


@pytest.mark.gpu
def test_bench_rccl_branch_runs_on_one_rank():
    """The `nccl` (= RCCL) branch of bench.py and shard.gather_records, executed for real: one rank on this GPU with
    --force-dist, so the communicator is created, the barriers, the MAX all-reduce of the timing and the all-gather of the
    per-clip records all run through RCCL.  The 1 -> 8 curve is the driver's to measure (concat.sh:248
    is the model: independent clips, one per GPU); this only proves the branch executes and leaves stdout clean."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--workload", "1080p",
                        "--steps", "2", "--batch", "8", "--preroll", "40", "--fixed-preroll", "--ring", "16", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout   # RCCL's banner goes to stderr, stdout is the one JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["parity_check"] == "ok" and line["config"]["clips"] == 1 and line["value"] > 0
    assert line["collectives"] == "nccl"


@pytest.mark.gpu
def test_bench_line_of_the_plane_wise_outputs():
    """bench.py --out-format nv12-planar / p010-planar (short runs): one JSON line, the roofline object priced on the PLANE-WISE kernel's
    algorithmic bytes (source planes + output planes, no BGR frame), the emitted frame checked against the plane-wise checker, per-step
    durations and the host cost in the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for workload, fmt, bytes_per_launch in (("1080p", "nv12-planar", 1920 * 1080 * 3 // 2 + 1759 * 998 + 2 * 880 * 499),
                                            ("4k-p010", "p010-planar", 2 * (3840 * 2160 * 3 // 2 + 3524 * 1999 + 2 * 1762 * 1000))):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--out-format", fmt, "--steps", "3", "--warmup", "1",
                            "--batch", "16", "--preroll", "80", "--fixed-preroll", "--ring", "16", "--no-cpu-baseline"],
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
        line = json.loads(lines[0])
        assert line["parity_check"] == "ok" and line["value"] > 0 and line["unit"] == "frames/s", line
        roof = line["roofline"]
        assert "k_warp_planar" in roof["kernel"] and roof["bound"] == "hbm" and roof["unit"] == "GB/s", roof
        assert roof["algorithmic_bytes_per_launch"] == bytes_per_launch, (roof["algorithmic_bytes_per_launch"], bytes_per_launch)
        assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
        assert len(line["step_ms"]["all"]) == 3 and line["step_ms"]["min"] <= line["step_ms"]["median"] <= line["step_ms"]["max"]
        assert line["host"]["cpu_seconds_per_1000_frames"] > 0 and line["host"]["pinned_cpus"] >= 1
