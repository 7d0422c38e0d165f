"""TEST INFRASTRUCTURE — ctypes binding of the C oracle (oracle/liboracle.so, see oracle/c/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(rs-face-detection-tflite_amd/) never does.  Pinned by the reference's own rendered PNGs, pixel for pixel, on man.jpg
(tests/test_pins.py); what that pin covers and what it does not is stated in oracle/c/oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FD_FRONT, FD_BACK, FD_SHORT, FD_FULL, FD_FULL_SPARSE = 0, 1, 2, 3, 4


class Detection(C.Structure):
    _fields_ = [("data", C.c_float * 16), ("score", C.c_float)]


class Rect(C.Structure):
    _fields_ = [("x_center", C.c_double), ("y_center", C.c_double), ("width", C.c_double), ("height", C.c_double),
                ("rotation", C.c_double), ("normalized", C.c_int)]


def build(force=False):
    """Compile oracle/liboracle.so with the committed Makefile (gcc; a few seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = build()
    try:
        L = C.CDLL(so)
    except OSError:
        so = build(force=True)
        L = C.CDLL(so)
    fp, ip, dp, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p
    L.orc_last_error.restype = C.c_char_p
    L.orc_model_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.orc_model_load_bytes.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
    L.orc_model_free.argtypes = [vp]
    L.orc_model_input_dims.argtypes = [vp, ip]
    L.orc_model_num_outputs.argtypes = [vp]
    L.orc_model_output_dims.argtypes = [vp, C.c_int, ip]
    L.orc_model_output_elems.argtypes = [vp, C.c_int]
    L.orc_model_output_elems.restype = C.c_size_t
    L.orc_model_run.argtypes = [vp, fp, C.c_int, C.POINTER(fp), C.c_int]
    L.orc_model_run_tensor.argtypes = [vp, fp, C.c_int, fp, C.c_size_t]
    L.orc_model_run_tensor.restype = C.c_long
    L.orc_ssd_anchors.argtypes = [C.c_int, fp, C.c_int]
    L.orc_fd_input_size.argtypes = [C.c_int]
    L.orc_decode_boxes.argtypes = [fp, fp, C.c_int, C.c_float, fp]
    L.orc_sigmoid_scores.argtypes = [fp, C.c_int, fp]
    L.orc_convert_to_detections.argtypes = [fp, fp, C.c_int, C.POINTER(Detection)]
    L.orc_weighted_nms.argtypes = [C.POINTER(Detection), C.c_int, C.c_float, C.c_int, C.c_float, C.POINTER(Detection)]
    L.orc_plain_nms.argtypes = L.orc_weighted_nms.argtypes
    L.orc_letterbox_removal.argtypes = [C.POINTER(Detection), C.c_int, dp]
    L.orc_fd_postprocess.argtypes = [fp, fp, fp, C.c_int, C.c_float, dp, C.POINTER(Detection), C.c_int]
    L.orc_project_landmarks.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, C.POINTER(Rect), C.c_int, dp]
    L.orc_face_flag_passes.argtypes = [C.c_float]
    L.orc_bbox_to_roi.argtypes = [dp, C.c_int, C.c_int, dp, C.c_double, C.c_double, C.c_int, C.POINTER(Rect)]
    L.orc_face_detection_to_roi.argtypes = [C.POINTER(Detection), C.c_int, C.c_int, C.POINTER(Rect)]
    L.orc_iris_rois_from_face_landmarks.argtypes = [dp, C.c_int, C.c_int, C.POINTER(Rect), C.POINTER(Rect)]
    L.orc_image_to_tensor.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(Rect), C.c_int, C.c_int, C.c_int,
                                      C.c_double, C.c_double, C.c_int, fp, dp]
    _LIB = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def dets_to_array(dets, n):
    """-> float32 [n,17] (16 data + score)."""
    out = np.zeros((n, 17), np.float32)
    for i in range(n):
        out[i, :16] = np.frombuffer(dets[i].data, np.float32, 16)
        out[i, 16] = dets[i].score
    return out


def array_to_dets(arr):
    arr = np.asarray(arr, np.float32).reshape(-1, 17)
    d = (Detection * max(1, len(arr)))()
    for i, r in enumerate(arr):
        d[i].data[:] = [float(v) for v in r[:16]]
        d[i].score = float(r[16])
    return d


class Model:
    """Oracle model handle: FlatBufferModel + interpreter of the reference (face_detection.rs:188,207-257)."""

    def __init__(self, path):
        self.L = lib()
        self.h = C.c_void_p()
        if self.L.orc_model_load(os.fsencode(path), C.byref(self.h)) != 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        d = (C.c_int * 4)()
        self.L.orc_model_input_dims(self.h, d)
        self.input_dims = list(d)
        self.num_outputs = self.L.orc_model_num_outputs(self.h)
        self.output_dims = []
        self.output_elems = []
        for i in range(self.num_outputs):
            r = self.L.orc_model_output_dims(self.h, i, d)
            self.output_dims.append(list(d)[:r])
            self.output_elems.append(self.L.orc_model_output_elems(self.h, i))

    def __del__(self):
        try:
            if self.h:
                self.L.orc_model_free(self.h)
                self.h = None
        except Exception:
            pass

    def run(self, x, nthreads=1):
        x = np.ascontiguousarray(x, np.float32)
        assert list(x.shape[1:]) == self.input_dims[1:], (x.shape, self.input_dims)
        B = x.shape[0]
        outs = [np.empty([B] + dims[1:], np.float32) for dims in self.output_dims]
        ptrs = (C.POINTER(C.c_float) * self.num_outputs)(*[_fp(o) for o in outs])
        if self.L.orc_model_run(self.h, _fp(x), B, ptrs, nthreads) != 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return outs

    def tensor(self, x, index, cap=1 << 24):
        x = np.ascontiguousarray(x, np.float32)
        dst = np.empty(cap, np.float32)
        n = self.L.orc_model_run_tensor(self.h, _fp(x), index, _fp(dst), cap)
        if n < 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return dst[:n].copy()


def ssd_anchors(kind):
    L = lib()
    n = L.orc_ssd_anchors(kind, None, 0)
    a = np.zeros((n, 2), np.float32)
    L.orc_ssd_anchors(kind, _fp(a), n)
    return a


def fd_postprocess(raw_boxes, raw_scores, anchors, scale, padding=(0., 0., 0., 0.)):
    """One frame: raw_boxes [N,16], raw_scores [N] -> float32 [k,17]."""
    L = lib()
    rb = np.ascontiguousarray(raw_boxes, np.float32).reshape(-1, 16)
    rs = np.ascontiguousarray(raw_scores, np.float32).reshape(-1)
    an = np.ascontiguousarray(anchors, np.float32)
    n = rb.shape[0]
    out = (Detection * n)()
    pad = np.asarray(padding, np.float64)
    k = L.orc_fd_postprocess(_fp(rb), _fp(rs), _fp(an), n, float(scale), _dp(pad), out, n)
    if k < 0:
        raise RuntimeError("letterbox scale too small")
    return dets_to_array(out, k)


def project_landmarks(raw, tensor_size, image_size, padding=(0., 0., 0., 0.), roi=None, flip=False):
    L = lib()
    raw = np.ascontiguousarray(raw, np.float32).reshape(-1)
    n = raw.size // 3
    out = np.zeros((n, 3), np.float64)
    pad = np.asarray(padding, np.float64)
    r = C.byref(roi) if roi is not None else None
    L.orc_project_landmarks(_fp(raw), n, tensor_size[0], tensor_size[1], image_size[0], image_size[1], _dp(pad), r,
                            int(flip), _dp(out))
    return out


def image_to_tensor(image, roi, out_size, keep_aspect, out_range, flip):
    """image: uint8 [H,W,3] RGB. Returns (tensor [out_h,out_w,3] f32, padding(4))."""
    L = lib()
    image = np.ascontiguousarray(image, np.uint8)
    H, W = image.shape[:2]
    out = np.zeros((out_size[1], out_size[0], 3), np.float32)
    pad = np.zeros(4, np.float64)
    r = C.byref(roi) if roi is not None else None
    rc = L.orc_image_to_tensor(image.ctypes.data_as(C.POINTER(C.c_uint8)), W, H, r, out_size[0], out_size[1],
                               int(keep_aspect), out_range[0], out_range[1], int(flip), _fp(out), _dp(pad))
    if rc != 0:
        raise RuntimeError("image_to_tensor failed")
    return out, tuple(pad)


def face_detection_to_roi(det_row, image_size):
    L = lib()
    d = array_to_dets(np.asarray(det_row, np.float32).reshape(1, 17))
    r = Rect()
    if L.orc_face_detection_to_roi(d, image_size[0], image_size[1], C.byref(r)) != 0:
        raise RuntimeError("bbox must be normalized")
    return r


def iris_rois_from_face_landmarks(landmarks, image_size):
    L = lib()
    lm = np.ascontiguousarray(landmarks, np.float64)
    a, b = Rect(), Rect()
    if L.orc_iris_rois_from_face_landmarks(_dp(lm), image_size[0], image_size[1], C.byref(a), C.byref(b)) != 0:
        raise RuntimeError("bbox must be normalized")
    return a, b


# iris_landmark.rs:64-95: eye-contour landmark n refines face-mesh landmark EYE_TO_FACE[eye][n] (0 = left, 1 = right)
EYE_TO_FACE = (
    (33, 7, 163, 144, 145, 153, 154, 155, 133, 246, 161, 160, 159, 158, 157, 173, 130, 25, 110, 24, 23, 22, 26, 112, 243, 247, 30, 29, 27, 28, 56, 190, 226, 31, 228, 229, 230, 231, 232, 233, 244, 113, 225, 224, 223, 222, 221, 189, 35, 124, 46, 53, 52, 65, 143, 111, 117, 118, 119, 120, 121, 128, 245, 156, 70, 63, 105, 66, 107, 55, 193),
    (263, 249, 390, 373, 374, 380, 381, 382, 362, 466, 388, 387, 386, 385, 384, 398, 359, 255, 339, 254, 253, 252, 256, 341, 463, 467, 260, 259, 257, 258, 286, 414, 446, 261, 448, 449, 450, 451, 452, 453, 464, 342, 445, 444, 443, 442, 441, 413, 265, 353, 276, 283, 282, 295, 372, 340, 346, 347, 348, 349, 350, 357, 465, 383, 300, 293, 334, 296, 336, 285, 417),
)


def update_face_landmarks_with_iris_results(face, left_contour, right_contour):
    """iris_landmark.rs:380-398 restated on arrays: face [468,3], contours [71,3] -> refined [468,3] (left loop, then right)."""
    face = np.asarray(face, np.float64)
    if face.shape[0] != 468:
        raise ValueError("unexpected number of items in face_landmarks")
    out = face.copy()
    for eye, contour in enumerate((left_contour, right_contour)):
        for n, point in enumerate(np.asarray(contour, np.float64)):
            out[EYE_TO_FACE[eye][n]] = point
    return out


def jpeg_decode_rgb(data: bytes):
    """convert_image_to_mat (utils.rs:8-21): JPEG bytes -> [H, W, 3] u8 RGB, by the C restatement (oracle/c/jpeg.c)."""
    L = lib()
    L.orc_jpeg_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_jpeg_decode_rgb.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
    w, h = C.c_int(), C.c_int()
    if L.orc_jpeg_info(data, len(data), C.byref(w), C.byref(h)) != 0:
        raise ValueError("not a JPEG with a frame header")
    out = np.zeros((h.value, w.value, 3), np.uint8)
    rc = L.orc_jpeg_decode_rgb(data, len(data), out.ctypes.data, w.value, h.value)
    if rc != 0:
        raise ValueError("jpeg decode failed (%d)" % rc)
    return out
