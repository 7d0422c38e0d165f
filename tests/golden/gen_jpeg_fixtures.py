"""Generates tests/golden/jpeg/*: small JPEG variants (sampling modes, restart intervals, greyscale, odd sizes, progressive) encoded with
Pillow from a crop of the reference's test_data/man.jpg, and jpeg_pins.json = SHA-256 of libjpeg-turbo's decode (Pillow's
decoder: libjpeg-turbo, 6.2 API — the library family behind cv::imdecode in the reference, utils.rs:13) of every fixture and
of the reference's three test JPEGs.  Run once in the build container: python tests/golden/gen_jpeg_fixtures.py"""
import hashlib, io, json, os
import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(HERE, "jpeg")
os.makedirs(out, exist_ok=True)
base = np.asarray(Image.open(os.path.join(HERE, "man.jpg")).convert("RGB"))
variants = {
    "c444.jpg": (base[40:137, 100:231], dict(subsampling=0, quality=87)),
    "c422.jpg": (base[40:137, 100:231], dict(subsampling=1, quality=87)),
    "c420.jpg": (base[40:137, 100:231], dict(subsampling=2, quality=87)),
    "c420_rst3.jpg": (base[40:137, 100:231], dict(subsampling=2, quality=70, restart_marker_blocks=3)),
    "c444_rstrow.jpg": (base[30:95, 150:283], dict(subsampling=0, quality=95, restart_marker_rows=1)),
    "c420_q30_optimized.jpg": (base[0:200, 120:433], dict(subsampling=2, quality=30, optimize=True)),
    "c420_tiny_3x5.jpg": (base[100:105, 200:203], dict(subsampling=2, quality=90)),
    "c420_17x16.jpg": (base[100:116, 200:217], dict(subsampling=2, quality=90)),
    "grey.jpg": (None, dict(quality=80)),
    # progressive (SOF2: spectral selection + successive approximation, several scans, per-scan Huffman tables): imdecode takes them
    "prog_c420.jpg": (base[40:137, 100:231], dict(subsampling=2, quality=80, progressive=True)),
    "prog_c444.jpg": (base[40:137, 100:231], dict(subsampling=0, quality=87, progressive=True)),
    "prog_c422_q92.jpg": (base[0:200, 120:433], dict(subsampling=1, quality=92, progressive=True)),
    "prog_c420_q30_17x16.jpg": (base[100:116, 200:217], dict(subsampling=2, quality=30, progressive=True, optimize=True)),
    "prog_grey.jpg": (None, dict(quality=80, progressive=True)),
}
pins = {"_decoder": "Pillow %s / libjpeg-turbo (jpeglib %s)" % (Image.__version__, features.version("jpg"))}
for name, (arr, kw) in variants.items():
    im = Image.fromarray(base[40:137, 100:231]).convert("L") if arr is None else Image.fromarray(np.ascontiguousarray(arr))
    bio = io.BytesIO()
    im.save(bio, "JPEG", **kw)
    open(os.path.join(out, name), "wb").write(bio.getvalue())
for path in sorted([os.path.join(out, n) for n in variants] + [os.path.join(HERE, n) for n in ("man.jpg", "russ_cox_1.jpg", "russ_cox_2.jpg")]):
    data = open(path, "rb").read()
    rgb = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    pins[os.path.relpath(path, HERE)] = {"shape": list(rgb.shape), "sha256": hashlib.sha256(rgb.tobytes()).hexdigest()}
# a coding process outside the supported subset (arithmetic coding: the frame marker of a baseline file rewritten to SOF9), must be refused
data = open(os.path.join(out, "c420.jpg"), "rb").read()
open(os.path.join(out, "arithmetic_unsupported.jpg"), "wb").write(data.replace(b"\xff\xc0", b"\xff\xc9", 1))
json.dump(pins, open(os.path.join(HERE, "jpeg_pins.json"), "w"), indent=1, sort_keys=True)
print(len(pins) - 1, "pins written")
