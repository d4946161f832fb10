"""vstab_gpmf_parse_gyro: GPMF payload -> gyro samples, the reader opencv/gpmf.cpp:33-114 sketches (commented out, with gpmf-parser) and
AvFrameSourceFileVaapi.cpp:121-123 leaves as a TODO.  gpmf-parser is not in the image and the reference holds no payloads, so the known
answers are hand-built here from GoPro's published KLV layout (key, type, size, repeat, big-endian data padded to four bytes)."""
import struct

import numpy as np
import pytest


def klv(key, typ, size, repeat, data):
    assert len(data) == size * repeat
    pad = (-len(data)) % 4
    return key.encode() + (bytes([typ]) if isinstance(typ, int) else typ.encode()) + bytes([size]) + struct.pack(">H", repeat) + data + b"\0" * pad


def nest(key, *items):
    body = b"".join(items)
    assert len(body) % 4 == 0 and len(body) < 65536
    return klv(key, 0, 1, len(body), body) if len(body) < 65536 else None


def gyro_stream(raw, scal, scal_type="s", gyro_type="s", extra=()):
    raw = np.asarray(raw)
    fmt = {"s": ">h", "S": ">H", "l": ">i", "L": ">I", "f": ">f", "d": ">d", "b": ">b"}
    sz = struct.calcsize(fmt[gyro_type])
    cast = float if gyro_type in "fd" else int
    data = b"".join(struct.pack(fmt[gyro_type], cast(v)) for v in raw.reshape(-1))
    scal = np.atleast_1d(scal)
    ssz = struct.calcsize(fmt[scal_type])
    sdata = b"".join(struct.pack(fmt[scal_type], (float if scal_type in "fd" else int)(v)) for v in scal)
    items = [klv("STNM", "c", 1, 4, b"Gyro"), klv("SIUN", "c", 5, 1, b"rad/s"), klv("TSMP", "L", 4, 1, struct.pack(">I", 1234)),
             klv("SCAL", scal_type, ssz, len(scal), sdata), *extra,
             klv("GYRO", gyro_type, 3 * sz, len(raw), data)]
    return nest("STRM", *items)


def test_known_answers_s16_with_scal_and_the_packet_span(vs):
    raw = np.array([[100, -200, 300], [0, 32767, -32768], [7, 8, 9], [-1, -2, -3]], np.int16)
    accl = nest("STRM", klv("SCAL", "s", 2, 1, struct.pack(">h", 418)), klv("ACCL", "s", 6, 2, struct.pack(">6h", 1, 2, 3, 4, 5, 6)))
    payload = nest("DEVC", klv("DVID", "L", 4, 1, struct.pack(">I", 1)), klv("DVNM", "c", 1, 6, b"Hero6 "), accl, gyro_stream(raw, 3755))
    got, n = vs.gpmf_parse_gyro(payload, 10.0, 0.2)
    assert n == 4 and got.shape == (4, 5)
    assert np.allclose(got[:, 0], 10.0 + 0.2 * np.arange(4) / 4, rtol=0, atol=1e-15) and np.allclose(got[:, 1] - got[:, 0], 0.05, rtol=0, atol=1e-15)   # gpmf.cpp:95-96
    assert np.array_equal(got[:, 2:], raw.astype(np.float64) / 3755.0)        # elements 0, 1, 2 -> roll, pitch, yaw (gpmf.cpp:97-99), raw / SCAL
    # a smaller buffer: the first samples, and the number needed
    part, n2 = vs.gpmf_parse_gyro(payload, 10.0, 0.2, cap=3)
    assert n2 == 4 and np.array_equal(part, got[:3])
    # the samples drive vstab_gyro_integrate: a constant rate about the optical axis (element 0 = Z = roll) turns the frame about z
    const = np.tile(np.array([[1000, 0, 0]], np.int16), (8, 1))
    s, _ = vs.gpmf_parse_gyro(nest("DEVC", gyro_stream(const, 1000)), 0.0, 1.0)
    Rd, _ = vs.gyro_integrate(s, -1.0, 0.0, 0.5, 0.5)
    a = -0.5    # body rate 1 rad/s for half a second, rate_scale -1: the scene turns the other way
    assert np.allclose(Rd, [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], atol=1e-12)


def test_scal_per_element_other_types_and_several_blocks(vs):
    raw = np.array([[10, 20, 30], [40, 50, 60]])
    p = nest("DEVC", gyro_stream(raw, [2, 4, 5], scal_type="l", gyro_type="l"))
    got, n = vs.gpmf_parse_gyro(p, 0.0, 1.0)
    assert n == 2 and np.array_equal(got[:, 2:], raw / np.array([2.0, 4.0, 5.0]))
    for gt, st, sc in (("f", "f", 0.5), ("d", "s", 3), ("b", "S", 7), ("S", "L", 9)):
        vals = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.float64)
        got, n = vs.gpmf_parse_gyro(nest("DEVC", gyro_stream(vals, sc, scal_type=st, gyro_type=gt)), 1.0, 3.0)
        assert n == 3 and np.array_equal(got[:, 2:], vals / sc), (gt, st)
    # a stream without SCAL: raw values; a five-element SCAL of another stream does not leak into a gyro block that follows it
    bare = nest("STRM", klv("GYRO", "s", 6, 1, struct.pack(">3h", 5, 6, 7)))
    gps = nest("STRM", klv("SCAL", "l", 4, 5, struct.pack(">5i", 10, 10, 10, 10, 10)), klv("GPS5", "l", 20, 1, struct.pack(">5i", 1, 2, 3, 4, 5)))
    got, n = vs.gpmf_parse_gyro(nest("DEVC", gps, bare), 0.0, 1.0)
    assert n == 1 and np.array_equal(got[0, 2:], [5, 6, 7])
    # two devices with a gyro each: both blocks, in order, each dealt over the packet's span
    two = nest("DEVC", gyro_stream(raw, 1)) + nest("DEVC", gyro_stream(raw * 2, 1))
    got, n = vs.gpmf_parse_gyro(two, 0.0, 1.0)
    assert n == 4 and np.array_equal(got[:, 2:], np.vstack([raw, raw * 2])) and np.array_equal(got[:, 0], [0, 0.5, 0, 0.5])
    # no gyro at all, trailing zero padding, an empty payload
    assert vs.gpmf_parse_gyro(nest("DEVC", gps) + b"\0" * 16, 0.0, 1.0)[1] == 0
    assert vs.gpmf_parse_gyro(b"\0\0\0\0", 0.0, 1.0)[1] == 0


def test_malformed_payloads_are_refused_not_read_past(vs):
    raw = np.array([[1, 2, 3], [4, 5, 6]], np.int16)
    good = nest("DEVC", gyro_stream(raw, 100))
    assert vs.gpmf_parse_gyro(good, 0.0, 1.0)[1] == 2
    bad = [
        good[:-4],                                                                   # the last item runs past the payload
        good[:11],                                                                   # a truncated header
        nest("DEVC", nest("STRM", klv("GYRO", "s", 4, 3, struct.pack(">6h", *range(6))))),     # two elements per sample (gpmf.cpp:88-92)
        nest("DEVC", nest("STRM", klv("GYRO", "c", 6, 1, b"abcdef"))),                          # not a number type
        nest("DEVC", nest("STRM", klv("SCAL", "s", 2, 1, struct.pack(">h", 0)), klv("GYRO", "s", 6, 1, struct.pack(">3h", 1, 2, 3)))),   # SCAL 0
        b"DEVC\0\x01\xff\xf0" + b"\0" * 64,                                          # a container that claims 65520 bytes
    ]
    deep = klv("GYRO", "s", 6, 1, struct.pack(">3h", 1, 2, 3))
    for _ in range(12):
        deep = nest("STRM", deep)
    bad.append(deep)                                                                 # nesting deeper than eight levels
    for k, b in enumerate(bad):
        with pytest.raises(vs.VstabError) as e:
            vs.gpmf_parse_gyro(b, 0.0, 1.0)
        assert e.value.status == vs.ERR_INVALID, k
    for ts, dur in ((0.0, -1.0), (float("nan"), 1.0), (0.0, float("inf")), (float("inf"), 1.0)):   # the packet's span has to be finite, the duration >= 0
        with pytest.raises(vs.VstabError):
            vs.gpmf_parse_gyro(good, ts, dur)
    for scal in (float("nan"), float("inf")):                                       # a divisor that is not a finite number
        with pytest.raises(vs.VstabError):
            vs.gpmf_parse_gyro(nest("DEVC", gyro_stream(raw, scal, scal_type="f")), 0.0, 1.0)
    # a sweep of damaged payloads: every prefix, and every single-byte corruption of the header bytes -- each either parses or is
    # refused; nothing crashes and nothing is read beyond the buffer (the sanitizer build runs this test too: tools/run_sanitized_tests.sh)
    rng = np.random.default_rng(3)
    for cut in range(len(good)):
        try:
            vs.gpmf_parse_gyro(good[:cut] if cut else b"\0", 0.0, 1.0)
        except vs.VstabError as e:
            assert e.status == vs.ERR_INVALID
    for _ in range(3000):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        try:
            got, n = vs.gpmf_parse_gyro(bytes(b), 0.0, 1.0)
            assert n >= 0 and np.isfinite(got).all()   # (integer samples over a finite non-zero divisor: every field finite)
        except vs.VstabError as e:
            assert e.status == vs.ERR_INVALID
