"""JPEG -> RGB (`convert_image_to_mat`, /root/reference/src/face_detection_lite/utils.rs:8-21).

CPU part: the oracle (oracle/c/jpeg.c) is pinned bit-exactly against libjpeg-turbo's own decode — the committed SHA-256 pins
of tests/golden/jpeg_pins.json, and Pillow live when it is importable — on the reference's three test JPEGs and on fixtures
covering the sampling modes, restart intervals, greyscale and odd sizes.  GPU part: the product (host entropy decoding +
HIP sample arithmetic, through the C ABI) against the oracle, bit-exact."""
import hashlib
import io
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

PINS = json.load(open(os.path.join(GOLDEN, "jpeg_pins.json")))
FILES = sorted(k for k in PINS if not k.startswith("_"))


def _bytes(rel):
    return open(os.path.join(GOLDEN, rel), "rb").read()


@pytest.mark.parametrize("rel", FILES)
def test_oracle_matches_libjpeg_turbo(oracle, rel):
    data = _bytes(rel)
    rgb = oracle.jpeg_decode_rgb(data)
    assert list(rgb.shape) == PINS[rel]["shape"]
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == PINS[rel]["sha256"]
    try:
        from PIL import Image
    except ImportError:
        return
    np.testing.assert_array_equal(rgb, np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))


def test_headers_and_refusals_without_gpu(mi, oracle):
    assert mi.jpeg_info(_bytes("man.jpg")) == (540, 360)          # the reference's test image (lib.rs:23)
    assert mi.jpeg_info(_bytes("russ_cox_2.jpg")) == (200, 225)
    for bad in (b"", b"\x89PNG\r\n\x1a\n" + b"\0" * 64, _bytes("man.jpg")[:300], _bytes("jpeg/arithmetic_unsupported.jpg")):
        with pytest.raises(mi.MiError):
            mi.convert_image_to_mat(bad)                           # refused before any device work
    with pytest.raises(ValueError):
        oracle.jpeg_decode_rgb(_bytes("jpeg/arithmetic_unsupported.jpg"))
    if mi.device_count() == 0:
        with pytest.raises(mi.MiError) as e:                       # the sample arithmetic has no CPU fallback
            mi.convert_image_to_mat(_bytes("man.jpg"))
        assert e.value.code == -4


def _segment(marker, body):
    return bytes([0xFF, marker]) + (len(body) + 2).to_bytes(2, "big") + body


def test_malformed_huffman_tables_and_sizes_are_refused(mi, oracle):
    """Untrusted bytes: a DHT whose length counts over-subscribe the code space (libjpeg: "bad Huffman table") used to index
    the 512-entry look-up table far out of range; pictures larger than 2^27 pixels are refused before anything is allocated."""
    sof = _segment(0xC0, bytes([8, 0, 16, 0, 16, 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1]))

    def dht(counts):
        total = sum(counts)
        return _segment(0xC4, bytes([0x00]) + bytes(counts) + bytes(i % 251 for i in range(total)))

    bad = ([200] + [0] * 15,            # 200 codes of length 1
           [0, 5] + [0] * 14,           # 5 codes of length 2
           [2, 1] + [0] * 14,           # both 1-bit codes used, then one more
           [2] + [0] * 15,              # both 1-bit codes used: the second is all ones (jdhuff.c: "no code is allowed to be all ones")
           [1, 1, 2] + [0] * 13,        # complete tree: its last code is 111
           [1] * 15 + [250])            # the last length overflows 16 bits
    good = ([0] * 8 + [255] + [0] * 7,  # 255 codes of length 9 fit
            [1] * 16)
    for counts in bad:
        with pytest.raises(mi.MiError) as e:
            mi.jpeg_info(b"\xff\xd8" + dht(counts) + sof + b"\xff\xd9")      # headers only: walks the DHT before the frame header
        assert "Huffman" in str(e.value)
        with pytest.raises(mi.MiError) as e:
            mi.convert_image_to_mat(b"\xff\xd8" + sof + dht(counts) + b"\xff\xd9")
        assert "Huffman" in str(e.value)
        with pytest.raises(ValueError):
            oracle.jpeg_decode_rgb(b"\xff\xd8" + sof + dht(counts) + b"\xff\xd9")   # (no scan) the checker does not crash either
    for counts in good:
        assert mi.jpeg_info(b"\xff\xd8" + dht(counts) + sof + b"\xff\xd9") == (16, 16)
    huge = b"\xff\xd8" + _segment(0xC0, bytes([8, 0xFF, 0xFF, 0xFF, 0xFF, 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1])) + b"\xff\xd9"
    with pytest.raises(mi.MiError) as e:
        mi.jpeg_info(huge)
    assert "too large" in str(e.value)


PROGRESSIVE = [k for k in FILES if "prog_" in k]


def _scan_cuts(data):
    """Variants of a progressive file with whole scans missing (cut in front of the k-th SOS, EOI appended) and one cut inside the last
    scan's entropy-coded data."""
    sos = [i for i in range(len(data) - 1) if data[i] == 0xFF and data[i + 1] == 0xDA]
    cuts = [("scans<%d" % k, data[: sos[k]] + b"\xff\xd9") for k in range(1, len(sos))]
    last = sos[-1] + 2 + int.from_bytes(data[sos[-1] + 2: sos[-1] + 4], "big")
    cuts.append(("last scan cut", data[: last + (len(data) - last) // 2]))
    return cuts


@pytest.mark.parametrize("rel", PROGRESSIVE)
def test_incomplete_progressive_streams_vs_libjpeg_turbo(oracle, rel):
    """libjpeg (hence cv::imdecode, utils.rs:10) decodes an INCOMPLETE progressive stream through inter-block smoothing
    (jdcoefct.c smoothing_ok / decompress_smooth_data) whenever every DC is known and one of the first AC coefficients is missing or
    not refined to its last bit.  That pass is not restated: such streams are REFUSED; everything the decoders accept must still be
    libjpeg-turbo's picture bit for bit (Pillow, truncated files allowed) — streams whose last scan is cut short included."""
    pil = pytest.importorskip("PIL.Image")
    from PIL import ImageFile
    accepted = refused = 0
    for what, cut in _scan_cuts(_bytes(rel)):
        ImageFile.LOAD_TRUNCATED_IMAGES = True
        try:
            ref = np.asarray(pil.open(io.BytesIO(cut)).convert("RGB"))
        finally:
            ImageFile.LOAD_TRUNCATED_IMAGES = False
        try:
            got = oracle.jpeg_decode_rgb(cut)
        except ValueError as e:
            assert "(-3)" in str(e), (rel, what, e)      # the incomplete-stream refusal, nothing else
            refused += 1
            continue
        np.testing.assert_array_equal(got, ref, err_msg="%s, %s" % (rel, what))
        accepted += 1
    assert accepted >= 1 and refused >= 1, (rel, accepted, refused)


@pytest.mark.gpu
@pytest.mark.parametrize("rel", PROGRESSIVE)
def test_gpu_incomplete_progressive_streams(mi, oracle, rel):
    """The product refuses exactly the streams the checker refuses (MI_EINVAL, "incomplete progressive stream": the caller falls back
    to imdecode) and agrees with it byte for byte on the others."""
    for what, cut in _scan_cuts(_bytes(rel)):
        try:
            want = oracle.jpeg_decode_rgb(cut)
        except ValueError:
            with pytest.raises(mi.MiError) as e:
                mi.convert_image_to_mat(cut)
            assert "incomplete progressive" in str(e.value) and e.value.code == -1, (rel, what)
            continue
        np.testing.assert_array_equal(mi.convert_image_to_mat(cut), want, err_msg="%s, %s" % (rel, what))


def test_progressive_allocation_guard_counts_one_bit_per_block(mi):
    """a progressive frame may spend a single bit per block (a DC-only first scan), so the allocation-bomb guard must not
    refuse a large flat progressive picture that libjpeg decodes: 32.5 MiB of coefficients backed by 1 bit per block pass the header
    check (they fail later, or not at all, for other reasons), the same header backed by 100 bytes does not."""
    W = H = 4128                                         # 516 x 516 blocks x 64 x 2 B = 32.5 MiB of coefficients
    sof = _segment(0xC2, bytes([8]) + H.to_bytes(2, "big") + W.to_bytes(2, "big") + bytes([1, 1, 0x11, 0]))
    dht = _segment(0xC4, bytes([0x00]) + bytes([1] + [0] * 15) + bytes([0]))       # one 1-bit code: DC difference 0
    sos = _segment(0xDA, bytes([1, 1, 0x00, 0, 0, 0x00]))
    blocks = (W // 8) * (H // 8)
    flat = b"\xff\xd8" + _segment(0xDB, bytes([0]) + bytes([16] * 64)) + sof + dht + sos + bytes(blocks // 8 + 1) + b"\xff\xd9"
    short = flat[: flat.index(b"\xff\xda") + 10 + 100] + b"\xff\xd9"
    for data, needle in ((short, "truncated stream"),):
        with pytest.raises(mi.MiError) as e:
            mi.convert_image_to_mat(data)
        assert needle in str(e.value)
    with pytest.raises(mi.MiError) as e:                 # passes the guard; stops at the next check (no device here / DC-only stream)
        mi.convert_image_to_mat(flat)
    assert "truncated stream" not in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("rel", FILES)
def test_gpu_decode_is_bit_exact(mi, oracle, rel):
    data = _bytes(rel)
    want = oracle.jpeg_decode_rgb(data)
    got = mi.convert_image_to_mat(data)
    np.testing.assert_array_equal(got, want)
    assert hashlib.sha256(got.tobytes()).hexdigest() == PINS[rel]["sha256"]
    dev = mi.convert_image_to_mat(data, to_device=True)            # pixels stay in HBM
    np.testing.assert_array_equal(dev.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("rel", ["man.jpg", "jpeg/c420_rst3.jpg", "jpeg/c444_rstrow.jpg", "jpeg/grey.jpg"])
def test_short_entropy_segment_decodes_like_libjpeg(mi, oracle, rel):
    """A picture cut off inside its entropy-coded data is not an error for libjpeg (hence for cv::imdecode, utils.rs:10): the MCU in
    which the data runs out is finished on padding zeros, the MCUs behind it stay zero = grey (jdhuff.c: insufficient_data), and with
    restart markers decoding resumes at the next marker found.  Product and checker agree byte for byte on cut and on holed files;
    only a header that promises more than 32 MiB of coefficients the remaining bytes cannot hold is refused (allocation bomb)."""
    data = _bytes(rel)
    full = oracle.jpeg_decode_rgb(data)
    sos = data.index(b"\xff\xda")
    ecs = sos + 2 + int.from_bytes(data[sos + 2: sos + 4], "big")      # first byte of the entropy-coded segment
    for frac in (0.35, 0.6, 0.9):
        cut = data[: ecs + int((len(data) - ecs) * frac)]
        want = oracle.jpeg_decode_rgb(cut)
        got = mi.convert_image_to_mat(cut)
        np.testing.assert_array_equal(got, want)
        assert got.shape == full.shape and not np.array_equal(got, full)
        H = got.shape[0]
        if rel == "man.jpg":
            assert np.array_equal(got[: H // 8], full[: H // 8])     # the rows decoded before the cut are the picture's
            assert (got[-4:] == got[-1, -1]).all()                    # the tail is uniform
    # a hole (bytes zeroed) in the middle of a file with restart markers: decoding picks up again behind it
    if "rst" in rel:
        holed = bytearray(data)
        a = ecs + (len(data) - ecs) // 2
        holed[a: a + min(40, (len(data) - a) // 2)] = bytes(min(40, (len(data) - a) // 2))
        np.testing.assert_array_equal(mi.convert_image_to_mat(bytes(holed)), oracle.jpeg_decode_rgb(bytes(holed)))
    bomb = b"\xff\xd8" + _segment(0xC0, bytes([8, 0x2E, 0xE0, 0x2E, 0xE0, 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1])) + data[2:400]
    with pytest.raises(mi.MiError):
        mi.convert_image_to_mat(bomb)


@pytest.mark.gpu
def test_bytes_to_detections_like_the_reference_test(mi):
    """lib.rs:18-29: include_bytes!(man.jpg) -> convert_image_to_mat -> FaceDetection::infer; same pin as the rendered bbox."""
    image = mi.convert_image_to_mat(_bytes("man.jpg"))
    faces = mi.FaceDetection(mi.FaceDetectionModel.BackCamera).infer(image, None)
    assert len(faces) == 1
    xmin, ymin, xmax, ymax = faces[0].bbox()
    H, W = image.shape[:2]
    assert int(xmin * W) == 195 and int(ymin * H) == 74 and int((xmax - xmin) * W) == 139 and int((ymax - ymin) * H) == 139


@pytest.mark.gpu
def test_streamed_jpeg_entries_equal_decode_then_infer(mi):
    """mi_fd_submit_jpeg / mi_fd_collect_jpeg (utils.rs:8-21 + face_detection.rs:205 for a stream of encoded pictures, two slots): every picture's
    detections are bit-equal to convert_image_to_mat + FaceDetection::infer of the same bytes — baseline, progressive, 4:4:4 / 4:2:2 / 4:2:0,
    greyscale, restart intervals, the reference's three test pictures — whatever is in flight in the other slot; a slot must be collected before
    it is reused, a bad stream is refused by submit and leaves the slot free; a single launch that gives up (engine test hook: absent workgroups)
    is repeated on the batched plan at collect."""
    rels = [r for r in FILES if "unsupported" not in r]
    for kind in (mi.FaceDetectionModel.BackCamera, mi.FaceDetectionModel.Short):
        fd = mi.FaceDetection(kind)
        ref = mi.FaceDetection(kind)
        want = {}
        for r in rels:
            want[r] = ref.infer(mi.convert_image_to_mat(_bytes(r)), None)
        assert len(want["man.jpg"]) == 1

        def same(got, r):
            assert len(got) == len(want[r]), r
            for a, b in zip(got, want[r]):
                np.testing.assert_array_equal(a.data, b.data, err_msg=r)
                assert a.score == b.score

        order = rels + rels[::-1] + ["man.jpg"] * 5
        fd.submit_jpeg(0, _bytes(order[0]))
        for i in range(1, len(order)):
            fd.submit_jpeg(i & 1, _bytes(order[i]))
            dets, size = fd.collect_jpeg((i - 1) & 1, with_size=True)
            same(dets, order[i - 1])
            assert size == mi.jpeg_info(_bytes(order[i - 1]))
        same(fd.collect_jpeg((len(order) - 1) & 1), order[-1])
        # slot discipline and refusals
        fd.submit_jpeg(0, _bytes("man.jpg"))
        with pytest.raises(mi.MiError):
            fd.submit_jpeg(0, _bytes("man.jpg"))                   # not collected yet
        with pytest.raises(mi.MiError):
            fd.submit_jpeg(1, _bytes("man.jpg")[:300])             # truncated headers: refused, slot 1 stays free
        fd.submit_jpeg(1, _bytes("russ_cox_1.jpg"))
        same(fd.collect_jpeg(0), "man.jpg")
        same(fd.collect_jpeg(1), "russ_cox_1.jpg")
        with pytest.raises(mi.MiError):
            fd._jpeg_cap[0] = 64
            fd.collect_jpeg(0)                                     # nothing submitted
        # the other entries of the handle still work between streamed pictures
        fd.submit_jpeg(0, _bytes("man.jpg"))
        got_img = fd.infer(mi.convert_image_to_mat(_bytes("russ_cox_2.jpg")), None)
        same(got_img, "russ_cox_2.jpg")
        same(fd.collect_jpeg(0), "man.jpg")
        # a single launch that gives up: the picture is repeated on the batched plan when it is collected (tolerance between the two plans)
        fd.model.set_option("band_test_absent", 4)
        fd.submit_jpeg(0, _bytes("man.jpg"))
        fd.submit_jpeg(1, _bytes("russ_cox_1.jpg"))
        for slot, r in ((0, "man.jpg"), (1, "russ_cox_1.jpg")):
            got = fd.collect_jpeg(slot)
            assert len(got) == len(want[r])
            for a, b in zip(got, want[r]):
                np.testing.assert_allclose(a.data, b.data, atol=1e-4)
        fd.close()
        ref.close()


@pytest.mark.gpu
def test_streamed_jpeg_pipeline_equals_decode_then_pipeline_run(mi):
    """mi_pipeline_submit_jpeg / mi_pipeline_collect_jpeg: the whole flow of the reference's own test (lib.rs:18-40: bytes -> convert_image_to_mat ->
    FaceDetection::infer -> FaceLandmark::infer -> 2 x IrisLandmark::infer) for a stream of encoded pictures, two slots.  Every picture's results are
    bit-equal to convert_image_to_mat + mi_pipeline_run (batch 1, host memory) on the same bytes, whatever is in flight in the other slot and
    whatever the picture's size (the reference's three test pictures: 540x360, 200x133, 200x225); a single launch that gives up is repeated."""
    rels = ["man.jpg", "russ_cox_1.jpg", "russ_cox_2.jpg", "jpeg/prog_420.jpg" if "jpeg/prog_420.jpg" in FILES else "man.jpg"]
    for kind in (mi.FaceDetectionModel.BackCamera, mi.FaceDetectionModel.Short):
        pipe = mi.Pipeline(kind)
        ref = mi.Pipeline(kind)
        want = {}
        for r in rels:
            img = mi.convert_image_to_mat(_bytes(r))
            want[r] = ref.run(np.ascontiguousarray(img[None]))
            want[r]["size"] = (img.shape[1], img.shape[0])
        assert want["man.jpg"]["face_counts"][0] >= 1 and want["man.jpg"]["present"][0] == 1

        def same(got, r, exact=True):
            assert got["size"] == want[r]["size"], r
            for k in ("faces", "face_counts", "landmarks", "present", "eyes"):
                if exact:
                    np.testing.assert_array_equal(got[k], want[r][k], err_msg="%s %s" % (r, k))
                else:
                    np.testing.assert_allclose(got[k], want[r][k], atol=2e-4, err_msg="%s %s" % (r, k))

        order = rels + rels[::-1] + ["man.jpg"] * 4
        pipe.submit_jpeg(0, _bytes(order[0]))
        for i in range(1, len(order)):
            pipe.submit_jpeg(i & 1, _bytes(order[i]))
            same(pipe.collect_jpeg((i - 1) & 1), order[i - 1])
        same(pipe.collect_jpeg((len(order) - 1) & 1), order[-1])
        with pytest.raises(mi.MiError):
            pipe.collect_jpeg(0)                                       # nothing submitted
        pipe.submit_jpeg(0, _bytes("man.jpg"))
        with pytest.raises(mi.MiError):
            pipe.submit_jpeg(0, _bytes("man.jpg"))                     # not collected yet
        with pytest.raises(mi.MiError):
            pipe.submit_jpeg(1, b"\xff\xd8\xff\xe0junk")               # not a picture: refused, slot 1 stays free
        same(pipe.collect_jpeg(0), "man.jpg")
        # mi_pipeline_run between two streamed pictures
        pipe.submit_jpeg(1, _bytes("russ_cox_1.jpg"))
        img = mi.convert_image_to_mat(_bytes("man.jpg"))
        got = pipe.run(np.ascontiguousarray(img[None]))
        for k in ("faces", "face_counts", "landmarks", "present", "eyes"):
            np.testing.assert_array_equal(got[k], want["man.jpg"][k])
        same(pipe.collect_jpeg(1), "russ_cox_1.jpg")
        # single launches that give up (absent workgroups in the detector's program): repeated on the batched plan at collect
        pipe.set_option("band_test_absent", 4)
        pipe.submit_jpeg(0, _bytes("man.jpg"))
        pipe.submit_jpeg(1, _bytes("russ_cox_2.jpg"))
        same(pipe.collect_jpeg(0), "man.jpg", exact=False)
        same(pipe.collect_jpeg(1), "russ_cox_2.jpg", exact=False)
        pipe.close()
        ref.close()
