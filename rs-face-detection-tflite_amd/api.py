"""Host-side mirror of the reference crate's public API over the C ABI (include/mi_face.h) via ctypes.

The reference is a Rust crate (`FaceDetection::{new,infer}`, `FaceLandmark::{new,infer}`, `IrisLandmark::{new,infer}`,
/root/reference/src/face_detection_lite/{face_detection,face_landmark,iris_landmark}.rs); there is no Rust toolchain in
this image, so the host side above the C ABI is mirrored here with the same names, argument meaning and error behaviour
(errors that the reference returns as `anyhow::Error` — or panics on — become `MiError`).  A Rust `extern "C"` shim with
identical signatures is listed in INTEGRATION.md.

No numerical work happens in this module: every result comes from the HIP kernels behind libmiface.so.  If that library
is missing, importing this module raises (there is no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass
from collections.abc import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIFACE_LIB") or os.path.join(_HERE, "libmiface.so")   # MIFACE_LIB: a development variant built by build.sh (MI_VARIANT)
REPO_ROOT = os.path.dirname(_HERE)
DEFAULT_MODEL_DIR = os.path.join(REPO_ROOT, "models")

MI_MEM_HOST, MI_MEM_DEVICE = 0, 1
NUM_FACE_LANDMARKS, NUM_EYE_LANDMARKS, NUM_IRIS_LANDMARKS = 468, 71, 5

EXPORTS = [
    "mi_last_error", "mi_device_count", "mi_version",
    "mi_model_load_file", "mi_model_load_bytes", "mi_model_free", "mi_model_input_dims", "mi_model_num_outputs",
    "mi_model_output_dims", "mi_model_output_elems", "mi_model_run", "mi_model_debug_tensor", "mi_model_describe",
    "mi_dist_broadcast_bytes", "mi_streams_create_distinct", "mi_streams_destroy", "mi_model_set_option", "mi_model_get_option", "mi_plan_describe", "mi_model_plan_stats", "mi_model_single_launch_workgroups", "mi_model_profile",
    "mi_fd_create", "mi_fd_create_from_bytes", "mi_fd_free", "mi_fd_model", "mi_fd_input_size", "mi_fd_num_anchors",
    "mi_fd_anchors", "mi_fd_infer_tensor", "mi_fd_postprocess", "mi_fd_infer_image", "mi_fd_infer_images", "mi_fd_submit_images",
    "mi_fd_collect", "mi_fd_submit_jpeg", "mi_fd_collect_jpeg", "mi_pipeline_submit_jpeg", "mi_pipeline_collect_jpeg", "mi_host_alloc", "mi_host_free",
    "mi_fl_create", "mi_fl_create_from_bytes", "mi_fl_free", "mi_fl_model", "mi_fl_infer_tensor", "mi_fl_infer_images", "mi_fl_submit_images", "mi_fl_collect", "mi_fl_infer_image",
    "mi_iris_create", "mi_iris_create_from_bytes", "mi_iris_free", "mi_iris_model", "mi_iris_infer_tensor", "mi_iris_infer_images",
    "mi_iris_infer_image",
    "mi_pipeline_create", "mi_pipeline_create_from_bytes", "mi_pipeline_model", "mi_pipeline_free", "mi_pipeline_set_option", "mi_pipeline_run",
    "mi_bbox_to_roi", "mi_bbox_from_landmarks", "mi_face_detection_to_roi", "mi_iris_roi_from_face_landmarks", "mi_update_face_landmarks_with_iris_results", "mi_image_to_tensor", "mi_jpeg_info", "mi_jpeg_decode_rgb",
]


class MiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mi_face error %d: %s" % (code, msg))
        self.code = code


class CDetection(C.Structure):
    _fields_ = [("data", C.c_float * 16), ("score", C.c_float)]


class Rect(C.Structure):
    """types.rs:24-36."""
    _fields_ = [("x_center", C.c_double), ("y_center", C.c_double), ("width", C.c_double), ("height", C.c_double),
                ("rotation", C.c_double), ("normalized", C.c_int)]

    def __repr__(self):
        return "Rect(x_center=%r, y_center=%r, width=%r, height=%r, rotation=%r, normalized=%r)" % (
            self.x_center, self.y_center, self.width, self.height, self.rotation, bool(self.normalized))


class CLandmark(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class FaceDetectionModel(enum.IntEnum):
    """face_detection.rs:117-123."""
    FrontCamera = 0
    BackCamera = 1
    Short = 2
    Full = 3
    FullSparse = 4


@dataclass
class Detection:
    """types.rs:189-246: data [8,2] f32 (bbox corners then 6 keypoints), score f32."""
    data: np.ndarray
    score: float

    def bbox(self):
        return tuple(float(v) for v in (self.data[0, 0], self.data[0, 1], self.data[1, 0], self.data[1, 1]))

    def keypoint(self, k):
        return float(self.data[k + 2, 0]), float(self.data[k + 2, 1])


@dataclass
class Landmark:
    x: float
    y: float
    z: float


class LandmarkList(Sequence):
    """Vec<Landmark> the way the C ABI filled it: one [n,3] f64 array (`.array`: x, y, z per row, types.rs:176-187).  It reads like the list of Landmark the
    reference returns (len, index, slice, iteration, ==); the Landmark objects are made on access, not per call — building 468 Python
    objects costs about as much as a third of the whole single-image mesh call (tools/latency_probe.py)."""
    __slots__ = ("array",)

    def __init__(self, array):
        self.array = array

    def __len__(self):
        return len(self.array)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return LandmarkList(self.array[i])
        x, y, z = self.array[i].tolist()
        return Landmark(x, y, z)

    def __iter__(self):
        it = iter(self.array.ravel().tolist())
        return map(Landmark, it, it, it)

    def __eq__(self, other):
        try:
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    def __repr__(self):
        return "LandmarkList(%d landmarks)" % len(self)


def _landmarks_from(carray, n):
    """n CLandmark structs -> LandmarkList over its own copy of their floats"""
    return LandmarkList(np.frombuffer(carray, np.float64, 3 * n).reshape(n, 3).copy())


@dataclass
class IrisResults:
    """iris_landmark.rs:115-129."""
    contour: list
    iris: list

    def eyeball_contour(self):
        return self.contour[:15]


_lib = None


def lib():
    """Load libmiface.so (built in-tree by build.sh / __graft_entry__.build()). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run rs-face-detection-tflite_amd/build.sh (hipcc, gfx950). "
                          "There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64; when both live in one process torch must be loaded first so that the
    # two share one HIP runtime (otherwise torch later reports "No HIP GPUs are available").  torch is only plumbing
    # here (device buffers, streams, torch.distributed) — set MIFACE_NO_TORCH=1 to skip it entirely.
    if not os.environ.get("MIFACE_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, fp, ip, dp = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.mi_last_error.restype = C.c_char_p
    L.mi_version.restype = C.c_char_p
    L.mi_model_load_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.mi_model_load_bytes.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(vp)]
    L.mi_model_free.argtypes = [vp]
    L.mi_model_free.restype = None
    L.mi_model_input_dims.argtypes = [vp, ip]
    L.mi_model_num_outputs.argtypes = [vp]
    L.mi_model_output_dims.argtypes = [vp, C.c_int, ip, ip]
    L.mi_model_output_elems.argtypes = [vp, C.c_int]
    L.mi_model_output_elems.restype = C.c_size_t
    L.mi_model_run.argtypes = [vp, vp, C.c_int, C.POINTER(vp), C.c_int, vp]
    L.mi_model_debug_tensor.argtypes = [vp, C.c_int, C.c_int, fp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mi_model_describe.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.mi_model_describe.restype = C.c_size_t
    L.mi_model_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.mi_model_get_option.argtypes = [vp, C.c_char_p, ip]
    L.mi_plan_describe.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t]
    L.mi_plan_describe.restype = C.c_size_t
    L.mi_model_plan_stats.argtypes = [vp, dp, dp, ip]
    L.mi_model_single_launch_workgroups.argtypes = [vp, C.c_int]
    L.mi_model_profile.argtypes = [vp, vp, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.mi_model_profile.restype = C.c_size_t
    L.mi_fd_create.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(vp)]
    L.mi_fd_create_from_bytes.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(vp)]
    L.mi_fd_free.argtypes = [vp]
    L.mi_fd_free.restype = None
    L.mi_fd_model.argtypes = [vp]
    L.mi_fd_model.restype = vp
    L.mi_fd_input_size.argtypes = [vp, ip, ip]
    L.mi_fd_num_anchors.argtypes = [vp]
    L.mi_fd_anchors.argtypes = [vp, fp, C.c_int]
    L.mi_fd_infer_tensor.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp]
    L.mi_fd_postprocess.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp]
    L.mi_fd_infer_images.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp]
    L.mi_fd_submit_images.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.mi_fd_collect.argtypes = [vp, C.c_int, vp, vp]
    L.mi_fd_submit_jpeg.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t, C.c_int]
    L.mi_fd_collect_jpeg.argtypes = [vp, C.c_int, vp, C.c_int, ip, ip, ip]
    L.mi_pipeline_submit_jpeg.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t]
    L.mi_pipeline_collect_jpeg.argtypes = [vp, C.c_int, vp, ip, vp, ip, vp, ip, ip]
    L.mi_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.mi_host_free.argtypes = [vp]
    L.mi_host_free.restype = None
    L.mi_fd_infer_image.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(Rect), C.POINTER(CDetection), C.c_int, ip]
    L.mi_fl_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.mi_fl_create_from_bytes.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(vp)]
    L.mi_fl_free.argtypes = [vp]
    L.mi_fl_free.restype = None
    L.mi_fl_model.argtypes = [vp]
    L.mi_fl_model.restype = vp
    L.mi_fl_infer_tensor.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp]
    L.mi_fl_infer_images.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, vp]
    L.mi_fl_submit_images.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.mi_fl_collect.argtypes = [vp, C.c_int, vp, vp, vp]
    L.mi_fl_infer_image.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(Rect), C.POINTER(CLandmark), C.c_int, ip]
    L.mi_iris_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.mi_iris_create_from_bytes.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(vp)]
    L.mi_iris_free.argtypes = [vp]
    L.mi_iris_free.restype = None
    L.mi_iris_model.argtypes = [vp]
    L.mi_iris_model.restype = vp
    L.mi_iris_infer_tensor.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp]
    L.mi_iris_infer_images.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, vp]
    L.mi_iris_infer_image.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(Rect), C.c_int, C.POINTER(CLandmark),
                                      C.POINTER(CLandmark)]
    L.mi_pipeline_create.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(vp)]
    L.mi_pipeline_create_from_bytes.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(vp)]
    L.mi_pipeline_model.argtypes = [vp, C.c_int]
    L.mi_pipeline_model.restype = vp
    L.mi_pipeline_free.argtypes = [vp]
    L.mi_pipeline_free.restype = None
    L.mi_pipeline_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.mi_pipeline_run.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp]
    L.mi_face_detection_to_roi.argtypes = [C.POINTER(CDetection), C.c_int, C.c_int, C.POINTER(Rect)]
    L.mi_iris_roi_from_face_landmarks.argtypes = [C.POINTER(CLandmark), C.c_int, C.c_int, C.POINTER(Rect), C.POINTER(Rect)]
    L.mi_update_face_landmarks_with_iris_results.argtypes = [C.POINTER(CLandmark)] * 4
    L.mi_bbox_to_roi.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int, C.POINTER(Rect)]
    L.mi_bbox_from_landmarks.argtypes = [C.POINTER(CLandmark), C.c_int, C.POINTER(C.c_double)]
    L.mi_jpeg_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mi_jpeg_decode_rgb.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_void_p]
    L.mi_image_to_tensor.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(Rect), C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_double, C.c_int, vp, dp, C.c_int, vp]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise MiError(rc, lib().mi_last_error().decode("utf-8", "replace"))


def device_count():
    return lib().mi_device_count()


def dist_broadcast_bytes(id_path, rank, world, root, data, nbytes=None, device=0, timeout_ms=60000) -> bytes:
    """mi_dist_broadcast_bytes: `data` (bytes) on the root rank, None elsewhere (then `nbytes` says how much to receive).  Returns the
    bytes every rank now holds.  Collective over `world` ranks, one process per GPU, RCCL directly (no torch.distributed)."""
    n = len(data) if data is not None else int(nbytes)
    buf = (C.c_uint8 * n)()
    if data is not None:
        C.memmove(buf, data, n)
    L = lib()
    L.mi_dist_broadcast_bytes.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    _check(L.mi_dist_broadcast_bytes(os.fsencode(id_path), rank, world, root, device, buf, n, timeout_ms))
    return bytes(buf)


def streams_create_distinct(n=2, device=0):
    """mi_streams_create_distinct: n hipStream_t handles (ints) that were tested to sit on distinct hardware queues — pass them as the
    `stream` argument of the batched entries of n handles to keep n batches in flight.  Free with streams_destroy."""
    L = lib()
    arr = (C.c_void_p * n)()
    L.mi_streams_create_distinct.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    _check(L.mi_streams_create_distinct(int(device), int(n), arr))
    return [int(v) for v in arr]


def streams_destroy(streams, device=0):
    L = lib()
    n = len(streams)
    arr = (C.c_void_p * n)(*[C.c_void_p(s) for s in streams])
    L.mi_streams_destroy.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    _check(L.mi_streams_destroy(int(device), n, arr))


def plan_describe(tflite_bytes: bytes, fuse_level=4) -> str:
    """Host-only lowering of a .tflite blob (no GPU needed)."""
    L = lib()
    n = L.mi_plan_describe(tflite_bytes, len(tflite_bytes), fuse_level, None, 0)
    if n == 0:
        raise MiError(-3, L.mi_last_error().decode())
    buf = C.create_string_buffer(n)
    L.mi_plan_describe(tflite_bytes, len(tflite_bytes), fuse_level, buf, n)
    return buf.value.decode()


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _device_ready(x, device, dims=None, dtype="float32"):
    """Checks made before a torch CUDA tensor's raw pointer is handed to the kernels: dtype, device ordinal and trailing
    shape.  Then orders the call after the work already queued by torch: the library launches on the handle's own
    (non-blocking) stream unless the caller passes one, and that stream has no ordering against torch's current stream, so
    everything torch queued so far (the producer of `x`, the zero fills of the outputs allocated next to it) is waited
    for here.  Callers that pass their own `stream` and pre-allocated outputs skip nothing by this: it is one host sync."""
    import torch
    if str(x.dtype) != "torch." + dtype:
        raise ValueError("expected a %s tensor, got %s" % (dtype, x.dtype))
    if not x.is_cuda or (x.device.index or 0) != device:
        raise ValueError("tensor lives on %s but the handle is bound to cuda:%d" % (x.device, device))
    if dims is not None and list(x.shape[1:]) != list(dims):
        raise ValueError("expected frames of shape %s, got %s" % (list(dims), list(x.shape[1:])))
    torch.cuda.current_stream(x.device).synchronize()


def _ptr(x):
    """(pointer, mem) of a numpy array (host) or torch CUDA tensor (device)."""
    if _is_torch(x):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return C.c_void_p(x.data_ptr()), (MI_MEM_DEVICE if x.is_cuda else MI_MEM_HOST)
    return C.c_void_p(x.ctypes.data), MI_MEM_HOST


class Model:
    """L0 engine handle (replaces the `tflite` crate's FlatBufferModel + Interpreter)."""

    def __init__(self, path=None, device=0, handle=None, owner=None):
        self.L = lib()
        self._owner = owner
        self.device = device
        if handle is not None:
            self.h = C.c_void_p(handle)
        else:
            self.h = C.c_void_p()
            _check(self.L.mi_model_load_file(os.fsencode(path), device, C.byref(self.h)))
        d = (C.c_int * 4)()
        _check(self.L.mi_model_input_dims(self.h, d))
        self.input_dims = list(d)
        self.num_outputs = self.L.mi_model_num_outputs(self.h)
        self.output_dims, self.output_elems = [], []
        for i in range(self.num_outputs):
            r = C.c_int()
            _check(self.L.mi_model_output_dims(self.h, i, d, C.byref(r)))
            self.output_dims.append(list(d)[:r.value])
            self.output_elems.append(self.L.mi_model_output_elems(self.h, i))

    def close(self):
        if self._owner is None and getattr(self, "h", None):
            self.L.mi_model_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        _check(self.L.mi_model_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int(0)
        _check(self.L.mi_model_get_option(self.h, key.encode(), C.byref(v)))
        return v.value

    def describe(self):
        n = self.L.mi_model_describe(self.h, None, 0)
        buf = C.create_string_buffer(n)
        self.L.mi_model_describe(self.h, buf, n)
        return buf.value.decode()

    def plan_stats(self):
        b, m, n = C.c_double(), C.c_double(), C.c_int()
        _check(self.L.mi_model_plan_stats(self.h, C.byref(b), C.byref(m), C.byref(n)))
        return b.value, m.value, n.value

    def single_launch_workgroups(self, batch=1):
        """Workgroups a single-image call of `batch` frames occupies on the single-launch plan (0: the graph has none / batch too large)."""
        n = self.L.mi_model_single_launch_workgroups(self.h, int(batch))
        if n < 0:
            _check(n)
        return n

    def profile(self, x_device, reps=5):
        """Per-launch HIP-event timing (x_device: torch CUDA tensor). Returns a list of dicts."""
        import json
        p, mem = _ptr(x_device)
        if mem != MI_MEM_DEVICE:
            raise ValueError("profile() needs a device tensor")
        _device_ready(x_device, self.device, self.input_dims[1:])
        B = int(x_device.shape[0])
        cap = 1 << 20
        buf = C.create_string_buffer(cap)
        if self.L.mi_model_profile(self.h, p, B, reps, buf, cap) == 0:
            raise MiError(-1, self.L.mi_last_error().decode())
        return json.loads(buf.value.decode())

    def run(self, x, outs=None, stream=None):
        """x: numpy [B,H,W,C] f32 (host) or torch CUDA tensor. Returns list of outputs in the same memory space."""
        p, mem = _ptr(x)
        B = int(x.shape[0])
        if outs is None:
            if mem == MI_MEM_DEVICE:
                import torch
                outs = [torch.empty([B] + d[1:], dtype=torch.float32, device=x.device) for d in self.output_dims]
            else:
                x = np.ascontiguousarray(x, np.float32)
                p = C.c_void_p(x.ctypes.data)
                outs = [np.empty([B] + d[1:], np.float32) for d in self.output_dims]
        if mem == MI_MEM_DEVICE:
            _device_ready(x, self.device, self.input_dims[1:])
        ptrs = (C.c_void_p * self.num_outputs)(*[_ptr(o)[0] for o in outs])
        _check(self.L.mi_model_run(self.h, p, B, ptrs, mem, C.c_void_p(stream or 0)))
        return outs

    def debug_tensor(self, index, frame=0, cap=1 << 24):
        dst = np.empty(cap, np.float32)
        n = C.c_size_t()
        _check(self.L.mi_model_debug_tensor(self.h, index, frame, dst.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n)))
        return dst[:n.value].copy()


def _image_args(image):
    image = np.ascontiguousarray(image, np.uint8)
    if image.ndim != 3 or image.shape[2] != 3:
        raise ValueError("image must be uint8 [H,W,3] RGB")
    h, w = image.shape[:2]
    return image, w, h, image.strides[0]



def _frames_and_rois(frames, rois, items_per_frame, device):
    """(pointer, mem, B, H, W, stride, rois pointer, keep-alive) for the batched u8 entry points: frames uint8 [B,H,W,3] (numpy, or a
    contiguous torch CUDA tensor); rois None, a sequence of B * items_per_frame Rect (host frames) or a device tensor holding them."""
    p, mem = _ptr(frames)
    B, H, W = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    N = B * items_per_frame
    if rois is None and items_per_frame != 1:
        raise ValueError("several items per frame need their ROIs")
    if mem == MI_MEM_DEVICE:
        import torch
        if frames.dtype != torch.uint8 or not frames.is_contiguous() or frames.shape[3] != 3:
            raise ValueError("frames must be a contiguous uint8 [B,H,W,3] tensor")
        _device_ready(frames, device, None, dtype="uint8")
        rp = C.c_void_p(rois.data_ptr()) if rois is not None else None
        return p, mem, B, H, W, 3 * W, rp, frames
    frames = np.ascontiguousarray(frames, np.uint8)
    if frames.ndim != 4 or frames.shape[3] != 3:
        raise ValueError("frames must be uint8 [B,H,W,3]")
    rarr = None
    if rois is not None:
        if len(rois) != N:
            raise ValueError("expected %d rois" % N)
        rarr = rois if isinstance(rois, C.Array) else (Rect * N)(*rois)
    return C.c_void_p(frames.ctypes.data), mem, B, H, W, frames.strides[1], (C.cast(rarr, C.c_void_p) if rarr is not None else None), (frames, rarr)


class PinnedBuffer:
    """Page-locked host memory (mi_host_alloc) viewed as a numpy array; freed with the object."""

    def __init__(self, shape, dtype=np.uint8):
        self.L = lib()
        self.ptr = C.c_void_p()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        _check(self.L.mi_host_alloc(n, C.byref(self.ptr)))
        self.array = np.frombuffer((C.c_char * n).from_address(self.ptr.value), dtype=dtype).reshape(shape)

    def close(self):
        if getattr(self, "ptr", None) and self.ptr.value:
            self.array = None
            self.L.mi_host_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FaceDetection:
    """BlazeFace detector — mirrors face_detection.rs:146-267."""

    def __init__(self, model_type=FaceDetectionModel.FrontCamera, model_path=None, device=0, model_bytes=None):
        self.L = lib()
        self.h = C.c_void_p()
        # model_path is a DIRECTORY (face_detection.rs:157-161); default "./models" resolved against the repo here
        d = model_path if model_path is not None else DEFAULT_MODEL_DIR
        if model_bytes is not None:      # the frozen .tflite already in memory (e.g. after an RCCL broadcast)
            _check(self.L.mi_fd_create_from_bytes(int(model_type), model_bytes, len(model_bytes), device, C.byref(self.h)))
        else:
            _check(self.L.mi_fd_create(int(model_type), os.fsencode(d), device, C.byref(self.h)))
        w, h = C.c_int(), C.c_int()
        self.L.mi_fd_input_size(self.h, C.byref(w), C.byref(h))
        self.input_size = (w.value, h.value)
        self.num_anchors = self.L.mi_fd_num_anchors(self.h)
        self.device = device
        self.model = Model(handle=self.L.mi_fd_model(self.h), owner=self, device=device)

    def close(self):
        if getattr(self, "h", None):
            self.L.mi_fd_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def anchors(self):
        a = np.zeros((self.num_anchors, 2), np.float32)
        self.L.mi_fd_anchors(self.h, a.ctypes.data_as(C.POINTER(C.c_float)), self.num_anchors)
        return a

    def infer(self, image, roi=None, cap=256):
        """FaceDetection::infer(&Mat, Option<Rect>) -> Vec<Detection> (face_detection.rs:205-267)."""
        image, w, h, stride = _image_args(image)
        while True:
            out = (CDetection * cap)()
            n = C.c_int()
            _check(self.L.mi_fd_infer_image(self.h, C.c_void_p(image.ctypes.data), w, h, stride,
                                            C.byref(roi) if roi is not None else None, out, cap, C.byref(n)))
            if n.value <= cap:      # the reference returns every detection: never truncate silently
                break
            cap = n.value
        return [Detection(np.frombuffer(out[i].data, np.float32, 16).reshape(8, 2).copy(), float(out[i].score))
                for i in range(min(n.value, cap))]

    def infer_tensor(self, x, padding=None, cap=64, out=None, counts=None, stream=None):
        """Batched tensor-stage entry (configs 2/4). x: [B,H,W,3] f32 numpy or torch CUDA tensor in [-1,1].
        Returns (detections [B,cap,17] f32, counts [B] i32) in the same memory space as x."""
        p, mem = _ptr(x)
        B = int(x.shape[0])
        if mem == MI_MEM_DEVICE:
            import torch
            if out is None:
                out = torch.zeros((B, cap, 17), dtype=torch.float32, device=x.device)
            if counts is None:
                counts = torch.zeros((B,), dtype=torch.int32, device=x.device)
            pp = C.c_void_p(padding.data_ptr()) if padding is not None else None
            _device_ready(x, self.device, self.model.input_dims[1:])
        else:
            x = np.ascontiguousarray(x, np.float32)
            p = C.c_void_p(x.ctypes.data)
            out = np.zeros((B, cap, 17), np.float32) if out is None else out
            counts = np.zeros((B,), np.int32) if counts is None else counts
            if padding is not None:
                padding = np.ascontiguousarray(padding, np.float64).reshape(B, 4)
            pp = C.c_void_p(padding.ctypes.data) if padding is not None else None
        _check(self.L.mi_fd_infer_tensor(self.h, p, B, pp, _ptr(out)[0], cap, _ptr(counts)[0], mem, C.c_void_p(stream or 0)))
        return out, counts

    def infer_images(self, frames, rois=None, cap=64, out=None, counts=None, stream=None):
        """Batched FaceDetection::infer on u8 RGB frames [B,H,W,3] (numpy, or a torch CUDA uint8 tensor): device-side
        image_to_tensor + network + post-processing in one call.  rois: None or a sequence of B Rect (host frames only).
        Returns (detections [B,cap,17] f32, counts [B] i32) in the memory space of `frames`."""
        p, mem = _ptr(frames)
        B, H, W = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        if mem == MI_MEM_DEVICE:
            import torch
            if frames.dtype != torch.uint8 or not frames.is_contiguous() or frames.shape[3] != 3:
                raise ValueError("frames must be a contiguous uint8 [B,H,W,3] tensor")
            if rois is not None:
                raise ValueError("rois are host-side only here")
            stride = 3 * W
            out = torch.zeros((B, cap, 17), dtype=torch.float32, device=frames.device) if out is None else out
            counts = torch.zeros((B,), dtype=torch.int32, device=frames.device) if counts is None else counts
            rp = None
            _device_ready(frames, self.device, None, dtype="uint8")
        else:
            frames = np.ascontiguousarray(frames, np.uint8)
            if frames.ndim != 4 or frames.shape[3] != 3:
                raise ValueError("frames must be uint8 [B,H,W,3]")
            p, stride = C.c_void_p(frames.ctypes.data), frames.strides[1]
            out = np.zeros((B, cap, 17), np.float32) if out is None else out
            counts = np.zeros((B,), np.int32) if counts is None else counts
            rarr = (Rect * B)(*rois) if rois is not None else None
            rp = C.cast(rarr, C.c_void_p) if rarr is not None else None
        _check(self.L.mi_fd_infer_images(self.h, p, B, W, H, stride, rp, _ptr(out)[0], cap, _ptr(counts)[0], mem, C.c_void_p(stream or 0)))
        return out, counts

    def submit_images(self, slot, frames, cap=64):
        """Queues one batch of host frames (numpy uint8 [B,H,W,3]; pinned — `pinned_frames()` — for the copy to overlap the other
        slot's kernels) and returns at once; `collect(slot)` hands the results out."""
        if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[3] != 3 or not frames.flags["C_CONTIGUOUS"]:
            raise ValueError("frames must be a C-contiguous uint8 [B,H,W,3] array")
        B, H, W = frames.shape[:3]
        _check(self.L.mi_fd_submit_images(self.h, slot, C.c_void_p(frames.ctypes.data), B, W, H, frames.strides[1], cap))
        self._slot_shape = getattr(self, "_slot_shape", {})
        self._slot_shape[slot] = (B, cap, frames)   # keeps the frames alive until collect()

    def collect(self, slot):
        B, cap, _ = self._slot_shape.pop(slot)
        out = np.zeros((B, cap, 17), np.float32)
        counts = np.zeros((B,), np.int32)
        _check(self.L.mi_fd_collect(self.h, slot, C.c_void_p(out.ctypes.data), C.c_void_p(counts.ctypes.data)))
        return out, counts

    def submit_jpeg(self, slot, im_bytes, cap=64):
        """convert_image_to_mat + infer for a stream of encoded pictures (utils.rs:8-21, face_detection.rs:205): Huffman decoding happens in this
        call, on this thread, while the device still works on the other slot's picture; everything else is queued.  `collect_jpeg(slot)`
        returns the detections."""
        _check(self.L.mi_fd_submit_jpeg(self.h, slot, im_bytes, len(im_bytes), cap))
        self._jpeg_cap = getattr(self, "_jpeg_cap", {})
        self._jpeg_cap[slot] = cap

    def collect_jpeg(self, slot, with_size=False):
        cap = self._jpeg_cap.pop(slot)
        buf = (CDetection * cap)()
        n, w, h = C.c_int(0), C.c_int(0), C.c_int(0)
        _check(self.L.mi_fd_collect_jpeg(self.h, slot, buf, cap, C.byref(n), C.byref(w), C.byref(h)))
        dets = [Detection(np.frombuffer(buf[i].data, np.float32, 16).reshape(8, 2).copy(), float(buf[i].score)) for i in range(min(n.value, cap))]
        return (dets, (w.value, h.value)) if with_size else dets

    def postprocess(self, raw_boxes, raw_scores, padding=None, cap=64):
        """Post-network stage only (decode + sigmoid + weighted NMS + letterbox removal) on host arrays."""
        rb = np.ascontiguousarray(raw_boxes, np.float32).reshape(-1, self.num_anchors, 16)
        rs = np.ascontiguousarray(raw_scores, np.float32).reshape(-1, self.num_anchors)
        B = rb.shape[0]
        out = np.zeros((B, cap, 17), np.float32)
        counts = np.zeros((B,), np.int32)
        pp = None
        if padding is not None:
            padding = np.ascontiguousarray(padding, np.float64).reshape(B, 4)
            pp = C.c_void_p(padding.ctypes.data)
        _check(self.L.mi_fd_postprocess(self.h, C.c_void_p(rb.ctypes.data), C.c_void_p(rs.ctypes.data), B, pp,
                                        C.c_void_p(out.ctypes.data), cap, C.c_void_p(counts.ctypes.data), MI_MEM_HOST, None))
        return out, counts


class FaceLandmark:
    """468-point face mesh — mirrors face_landmark.rs:200-306."""

    def __init__(self, model_path=None, device=0, model_bytes=None):
        self.L = lib()
        self.h = C.c_void_p()
        p = model_path if model_path is not None else os.path.join(DEFAULT_MODEL_DIR, "face_landmark.tflite")
        if model_bytes is not None:
            _check(self.L.mi_fl_create_from_bytes(model_bytes, len(model_bytes), device, C.byref(self.h)))
        else:
            _check(self.L.mi_fl_create(os.fsencode(p), device, C.byref(self.h)))
        self.device = device
        self.model = Model(handle=self.L.mi_fl_model(self.h), owner=self, device=device)

    def close(self):
        if getattr(self, "h", None):
            self.L.mi_fl_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def infer(self, image, roi=None):
        """FaceLandmark::infer(&Mat, Option<Rect>) -> Vec<Landmark> (empty when the face flag fails)."""
        image, w, h, stride = _image_args(image)
        out = (CLandmark * NUM_FACE_LANDMARKS)()
        n = C.c_int()
        _check(self.L.mi_fl_infer_image(self.h, C.c_void_p(image.ctypes.data), w, h, stride,
                                        C.byref(roi) if roi is not None else None, out, NUM_FACE_LANDMARKS, C.byref(n)))
        return _landmarks_from(out, n.value)

    def infer_tensor(self, x, rois=None, image_sizes=None, stream=None):
        """x [B,192,192,3] in [0,1]. rois: list of Rect (host path) . Returns (landmarks [B,468,3], present [B], flags [B])."""
        p, mem = _ptr(x)
        B = int(x.shape[0])
        if mem == MI_MEM_DEVICE:
            import torch
            lm = torch.zeros((B, NUM_FACE_LANDMARKS, 3), dtype=torch.float32, device=x.device)
            present = torch.zeros((B,), dtype=torch.int32, device=x.device)
            flags = torch.zeros((B,), dtype=torch.float32, device=x.device)
            rp = C.c_void_p(rois.data_ptr()) if rois is not None else None
            sp = C.c_void_p(image_sizes.data_ptr()) if image_sizes is not None else None
            _device_ready(x, self.device, self.model.input_dims[1:])
        else:
            x = np.ascontiguousarray(x, np.float32)
            p = C.c_void_p(x.ctypes.data)
            lm = np.zeros((B, NUM_FACE_LANDMARKS, 3), np.float32)
            present = np.zeros((B,), np.int32)
            flags = np.zeros((B,), np.float32)
            rp = sp = None
            if rois is not None:
                arr = (Rect * B)(*rois)
                rp = C.cast(arr, C.c_void_p)
                image_sizes = np.ascontiguousarray(image_sizes, np.int32).reshape(B, 2)
                sp = C.c_void_p(image_sizes.ctypes.data)
        _check(self.L.mi_fl_infer_tensor(self.h, p, B, rp, sp, _ptr(lm)[0], _ptr(present)[0], _ptr(flags)[0], mem,
                                         C.c_void_p(stream or 0)))
        return lm, present, flags


    def infer_images(self, frames, rois=None, items_per_frame=1, stream=None):
        """Batched FaceLandmark::infer on u8 RGB frames [B,H,W,3] with one ROI per item (item i reads frame i // items_per_frame): device
        image_to_tensor + network + face flag + projection in one call.  Returns (landmarks [N,468,3], present [N], flags [N]) in the
        memory space of `frames`, N = B * items_per_frame."""
        p, mem, B, H, W, stride, rp, keep = _frames_and_rois(frames, rois, items_per_frame, self.device)
        N = B * items_per_frame
        if mem == MI_MEM_DEVICE:
            import torch
            lm = torch.zeros((N, NUM_FACE_LANDMARKS, 3), dtype=torch.float32, device=frames.device)
            present = torch.zeros((N,), dtype=torch.int32, device=frames.device)
            flags = torch.zeros((N,), dtype=torch.float32, device=frames.device)
        else:
            lm, present, flags = np.zeros((N, NUM_FACE_LANDMARKS, 3), np.float32), np.zeros((N,), np.int32), np.zeros((N,), np.float32)
        _check(self.L.mi_fl_infer_images(self.h, p, B, W, H, stride, rp, items_per_frame, _ptr(lm)[0], _ptr(present)[0], _ptr(flags)[0], mem,
                                         C.c_void_p(stream or 0)))
        return lm, present, flags


    def submit_images(self, slot, frames, rois=None, items_per_frame=1):
        """Queues one batch of host frames (numpy uint8 [B,H,W,3], pinned for the copy to overlap the other slot's kernels) with their ROIs
        and returns at once; `collect(slot)` hands the results out."""
        if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[3] != 3 or not frames.flags["C_CONTIGUOUS"]:
            raise ValueError("frames must be a C-contiguous uint8 [B,H,W,3] array")
        B, H, W = frames.shape[:3]
        N = B * items_per_frame
        rarr = None
        if rois is not None:
            rarr = rois if isinstance(rois, C.Array) else (Rect * N)(*rois)
            if len(rarr) != N:
                raise ValueError("expected %d rois" % N)
        _check(self.L.mi_fl_submit_images(self.h, slot, C.c_void_p(frames.ctypes.data), B, W, H, frames.strides[1],
                                          C.cast(rarr, C.c_void_p) if rarr is not None else None, items_per_frame))
        self._slot_shape = getattr(self, "_slot_shape", {})
        self._slot_shape[slot] = (N, frames, rarr)   # keeps frames and ROIs alive until collect()

    def collect(self, slot):
        N, _, _ = self._slot_shape.pop(slot)
        lm, present, flags = np.zeros((N, NUM_FACE_LANDMARKS, 3), np.float32), np.zeros((N,), np.int32), np.zeros((N,), np.float32)
        _check(self.L.mi_fl_collect(self.h, slot, C.c_void_p(lm.ctypes.data), C.c_void_p(present.ctypes.data), C.c_void_p(flags.ctypes.data)))
        return lm, present, flags


class IrisLandmark:
    """Iris / eye-contour model — mirrors iris_landmark.rs:130-248."""

    def __init__(self, model_path=None, device=0, model_bytes=None):
        self.L = lib()
        self.h = C.c_void_p()
        p = model_path if model_path is not None else os.path.join(DEFAULT_MODEL_DIR, "iris_landmark.tflite")
        if model_bytes is not None:
            _check(self.L.mi_iris_create_from_bytes(model_bytes, len(model_bytes), device, C.byref(self.h)))
        else:
            _check(self.L.mi_iris_create(os.fsencode(p), device, C.byref(self.h)))
        self.device = device
        self.model = Model(handle=self.L.mi_iris_model(self.h), owner=self, device=device)

    def close(self):
        if getattr(self, "h", None):
            self.L.mi_iris_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def infer(self, image, roi=None, is_right_eye=None):
        """IrisLandmark::infer(&Mat, Option<Rect>, Option<bool>) -> IrisResults."""
        image, w, h, stride = _image_args(image)
        c = (CLandmark * NUM_EYE_LANDMARKS)()
        i5 = (CLandmark * NUM_IRIS_LANDMARKS)()
        _check(self.L.mi_iris_infer_image(self.h, C.c_void_p(image.ctypes.data), w, h, stride,
                                          C.byref(roi) if roi is not None else None, int(bool(is_right_eye)), c, i5))
        return IrisResults(_landmarks_from(c, NUM_EYE_LANDMARKS), _landmarks_from(i5, NUM_IRIS_LANDMARKS))

    def infer_tensor(self, x, rois=None, image_sizes=None, padding=None, is_right_eye=None, stream=None):
        p, mem = _ptr(x)
        B = int(x.shape[0])
        if mem == MI_MEM_DEVICE:
            import torch
            contour = torch.zeros((B, NUM_EYE_LANDMARKS, 3), dtype=torch.float32, device=x.device)
            iris = torch.zeros((B, NUM_IRIS_LANDMARKS, 3), dtype=torch.float32, device=x.device)
            g = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
            rp, sp, pp, fp_ = g(rois), g(image_sizes), g(padding), g(is_right_eye)
            _device_ready(x, self.device, self.model.input_dims[1:])
        else:
            x = np.ascontiguousarray(x, np.float32)
            p = C.c_void_p(x.ctypes.data)
            contour = np.zeros((B, NUM_EYE_LANDMARKS, 3), np.float32)
            iris = np.zeros((B, NUM_IRIS_LANDMARKS, 3), np.float32)
            rp = sp = pp = fp_ = None
            keep = []
            if rois is not None:
                arr = (Rect * B)(*rois)
                keep.append(arr)
                rp = C.cast(arr, C.c_void_p)
                image_sizes = np.ascontiguousarray(image_sizes, np.int32).reshape(B, 2)
                sp = C.c_void_p(image_sizes.ctypes.data)
            if padding is not None:
                padding = np.ascontiguousarray(padding, np.float64).reshape(B, 4)
                pp = C.c_void_p(padding.ctypes.data)
            if is_right_eye is not None:
                is_right_eye = np.ascontiguousarray(is_right_eye, np.int32).reshape(B)
                fp_ = C.c_void_p(is_right_eye.ctypes.data)
        _check(self.L.mi_iris_infer_tensor(self.h, p, B, rp, sp, pp, fp_, _ptr(contour)[0], _ptr(iris)[0], mem,
                                           C.c_void_p(stream or 0)))
        return contour, iris


    def infer_images(self, frames, rois=None, is_right_eye=None, items_per_frame=1, stream=None):
        """Batched IrisLandmark::infer on u8 RGB frames [B,H,W,3] with one eye ROI (+ is_right_eye) per item (item i reads frame
        i // items_per_frame).  Returns (contour [N,71,3], iris [N,5,3]) in the memory space of `frames`."""
        p, mem, B, H, W, stride, rp, keep = _frames_and_rois(frames, rois, items_per_frame, self.device)
        N = B * items_per_frame
        fp_ = None
        if mem == MI_MEM_DEVICE:
            import torch
            contour = torch.zeros((N, NUM_EYE_LANDMARKS, 3), dtype=torch.float32, device=frames.device)
            iris = torch.zeros((N, NUM_IRIS_LANDMARKS, 3), dtype=torch.float32, device=frames.device)
            if is_right_eye is not None:
                fp_ = C.c_void_p(is_right_eye.data_ptr())
        else:
            contour, iris = np.zeros((N, NUM_EYE_LANDMARKS, 3), np.float32), np.zeros((N, NUM_IRIS_LANDMARKS, 3), np.float32)
            if is_right_eye is not None:
                is_right_eye = np.ascontiguousarray(is_right_eye, np.int32).reshape(N)
                fp_ = C.c_void_p(is_right_eye.ctypes.data)
        _check(self.L.mi_iris_infer_images(self.h, p, B, W, H, stride, rp, fp_, items_per_frame, _ptr(contour)[0], _ptr(iris)[0], mem,
                                           C.c_void_p(stream or 0)))
        return contour, iris


class Pipeline:
    """Batched detector -> mesh -> iris flow of lib.rs:18-40 / README.md:27-46, every stage on the GPU."""

    def __init__(self, model_type=FaceDetectionModel.BackCamera, model_dir=None, device=0, model_bytes=None):
        self.L = lib()
        self.h = C.c_void_p()
        d = model_dir if model_dir is not None else DEFAULT_MODEL_DIR
        self.device = device
        if model_bytes is not None:   # (detector, face_landmark, iris_landmark) .tflite blobs, e.g. received by RCCL broadcast
            fd_b, fl_b, ir_b = model_bytes
            _check(self.L.mi_pipeline_create_from_bytes(int(model_type), fd_b, len(fd_b), fl_b, len(fl_b), ir_b, len(ir_b), device, C.byref(self.h)))
        else:
            _check(self.L.mi_pipeline_create(int(model_type), os.fsencode(d), device, C.byref(self.h)))
        self.models = [Model(handle=self.L.mi_pipeline_model(self.h, k), owner=self, device=device) for k in range(3)]

    def close(self):
        if getattr(self, "h", None):
            self.L.mi_pipeline_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        _check(self.L.mi_pipeline_set_option(self.h, key.encode(), int(value)))

    def run(self, frames, stream=None):
        """frames: uint8 [B,H,W,3] RGB (numpy, or a torch CUDA tensor).
        Returns dict(faces [B,17], face_counts [B], landmarks [B,468,3], present [B], eyes [B,2,76,3])."""
        B, H, W = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        if _is_torch(frames) and frames.is_cuda:
            import torch
            mem, dev = MI_MEM_DEVICE, frames.device
            z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
            out = dict(faces=z((B, 17), torch.float32), face_counts=z((B,), torch.int32), landmarks=z((B, 468, 3), torch.float32),
                       present=z((B,), torch.int32), eyes=z((B, 2, 76, 3), torch.float32))
            p = C.c_void_p(frames.data_ptr())
            stride = int(frames.stride(1))
            if frames.dim() != 4 or frames.shape[3] != 3 or frames.stride(2) != 3 or frames.stride(3) != 1 or frames.stride(0) != stride * H:
                raise ValueError("frames must be [B,H,W,3] with dense pixels and frames stride*H bytes apart")
            _device_ready(frames, self.device, None, "uint8")
        else:
            frames = np.ascontiguousarray(frames, np.uint8)
            mem = MI_MEM_HOST
            out = dict(faces=np.zeros((B, 17), np.float32), face_counts=np.zeros((B,), np.int32), landmarks=np.zeros((B, 468, 3), np.float32),
                       present=np.zeros((B,), np.int32), eyes=np.zeros((B, 2, 76, 3), np.float32))
            p = C.c_void_p(frames.ctypes.data)
            stride = int(frames.strides[1])
        _check(self.L.mi_pipeline_run(self.h, p, B, W, H, stride, _ptr(out["faces"])[0], _ptr(out["face_counts"])[0],
                                      _ptr(out["landmarks"])[0], _ptr(out["present"])[0], _ptr(out["eyes"])[0], mem, C.c_void_p(stream or 0)))
        return out


def _pipeline_submit_jpeg(self, slot, im_bytes):
    """lib.rs:18-40 for a stream of encoded pictures: Huffman decoding of this picture happens in this call (on this thread, beside the device's work
    on the other slot's picture); the decoder's sample arithmetic and the three stages are queued.  `collect_jpeg(slot)` returns the results."""
    _check(self.L.mi_pipeline_submit_jpeg(self.h, slot, im_bytes, len(im_bytes)))


def _pipeline_collect_jpeg(self, slot):
    """-> dict(faces [1,17], face_counts [1], landmarks [1,468,3], present [1], eyes [1,2,76,3], size (w, h)) of the picture submitted to `slot`."""
    out = dict(faces=np.zeros((1, 17), np.float32), face_counts=np.zeros((1,), np.int32), landmarks=np.zeros((1, 468, 3), np.float32),
               present=np.zeros((1,), np.int32), eyes=np.zeros((1, 2, 76, 3), np.float32))
    w, h = C.c_int(0), C.c_int(0)
    _check(self.L.mi_pipeline_collect_jpeg(self.h, slot, _ptr(out["faces"])[0], C.cast(_ptr(out["face_counts"])[0], C.POINTER(C.c_int)), _ptr(out["landmarks"])[0],
                                           C.cast(_ptr(out["present"])[0], C.POINTER(C.c_int)), _ptr(out["eyes"])[0], C.byref(w), C.byref(h)))
    out["size"] = (w.value, h.value)
    return out


Pipeline.submit_jpeg = _pipeline_submit_jpeg
Pipeline.collect_jpeg = _pipeline_collect_jpeg


def face_detection_to_roi(face_detection: Detection, image_size, size_mode=None) -> Rect:
    """face_landmark.rs:180-198 (size_mode None = SquareLong, the only mode the reference's callers use)."""
    if size_mode not in (None, 1):
        raise ValueError("only SizeMode::SquareLong is exposed through the C ABI")
    d = CDetection()
    d.data[:] = [float(v) for v in np.asarray(face_detection.data, np.float32).reshape(16)]
    d.score = float(face_detection.score)
    r = Rect()
    _check(lib().mi_face_detection_to_roi(C.byref(d), int(image_size[0]), int(image_size[1]), C.byref(r)))
    return r


def iris_roi_from_face_landmarks(face_landmarks, image_size):
    """iris_landmark.rs:268-292 -> (left_eye_roi, right_eye_roi)."""
    if len(face_landmarks) < NUM_FACE_LANDMARKS:
        raise ValueError("expected 468 face landmarks")
    arr = (CLandmark * NUM_FACE_LANDMARKS)(*[CLandmark(l.x, l.y, l.z) for l in face_landmarks[:NUM_FACE_LANDMARKS]])
    a, b = Rect(), Rect()
    _check(lib().mi_iris_roi_from_face_landmarks(arr, int(image_size[0]), int(image_size[1]), C.byref(a), C.byref(b)))
    return a, b


def bbox_to_roi(bbox, image_size, rotation_keypoints=None, scale=(1.0, 1.0), size_mode=0) -> Rect:
    """transform.rs:44-109.  bbox (xmin, ymin, xmax, ymax) normalised; rotation_keypoints [(x0, y0), (x1, y1)] in pixels;
    size_mode 0 Default, 1 SquareLong, 2 SquareShort."""
    b = (C.c_double * 4)(*[float(v) for v in bbox])
    kp = None
    if rotation_keypoints is not None and len(rotation_keypoints) >= 2:   # fewer than two keypoints: rotation 0 (transform.rs:64-66)
        kp = (C.c_double * 4)(*[float(v) for p in rotation_keypoints[:2] for v in p])
    r = Rect()
    _check(lib().mi_bbox_to_roi(b, int(image_size[0]), int(image_size[1]), kp, float(scale[0]), float(scale[1]), int(size_mode), C.byref(r)))
    return r


def bbox_from_landmarks(landmarks):
    """transform.rs:146-165 -> (xmin, ymin, xmax, ymax)."""
    arr = (CLandmark * max(len(landmarks), 1))(*[CLandmark(l.x, l.y, l.z) for l in landmarks])
    out = (C.c_double * 4)()
    _check(lib().mi_bbox_from_landmarks(arr, len(landmarks), out))
    return tuple(out)


def jpeg_info(im_bytes: bytes):
    """(width, height) of a JPEG stream (headers only; works without a GPU)."""
    w, h = C.c_int(), C.c_int()
    _check(lib().mi_jpeg_info(im_bytes, len(im_bytes), C.byref(w), C.byref(h)))
    return w.value, h.value


def convert_image_to_mat(im_bytes: bytes, device=0, to_device=False):
    """utils.rs:8-21: encoded JPEG bytes -> 8UC3 RGB image [H, W, 3] u8 (numpy; a torch CUDA tensor with to_device=True,
    in which case the pixels never visit the host: Huffman decoding on the CPU, everything after it on the GPU)."""
    w, h = jpeg_info(im_bytes)
    ow, oh = C.c_int(), C.c_int()
    if to_device:
        import torch
        out = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:%d" % device)
        _check(lib().mi_jpeg_decode_rgb(device, im_bytes, len(im_bytes), C.c_void_p(out.data_ptr()), out.numel(), C.byref(ow), C.byref(oh), MI_MEM_DEVICE, None))
        return out
    out = np.empty((h, w, 3), np.uint8)
    _check(lib().mi_jpeg_decode_rgb(device, im_bytes, len(im_bytes), C.c_void_p(out.ctypes.data), out.size, C.byref(ow), C.byref(oh), MI_MEM_HOST, None))
    return out


def update_face_landmarks_with_iris_results(face_landmarks, iris_data_left, iris_data_right):
    """iris_landmark.rs:380-398: the eye-contour landmarks of both eyes replace the face-mesh points they refine."""
    if len(face_landmarks) != NUM_FACE_LANDMARKS:
        raise MiError(-1, "unexpected number of items in face_landmarks")
    pack = lambda v: (CLandmark * len(v))(*[CLandmark(l.x, l.y, l.z) for l in v])
    if len(iris_data_left.contour) != NUM_EYE_LANDMARKS or len(iris_data_right.contour) != NUM_EYE_LANDMARKS:
        raise MiError(-1, "expected 71 contour landmarks per eye")
    f, l, r = pack(face_landmarks), pack(iris_data_left.contour), pack(iris_data_right.contour)
    _check(lib().mi_update_face_landmarks_with_iris_results(f, l, r, f))
    return [Landmark(x.x, x.y, x.z) for x in f]


def image_to_tensor(image, roi=None, output_size=None, keep_aspect_ratio=False, output_range=(0., 1.), flip_horizontal=False,
                    device=0):
    """transform::image_to_tensor (transform.rs:188-309) on the GPU -> (tensor [h,w,3] f32 numpy, padding)."""
    image, w, h, stride = _image_args(image)
    ow, oh = output_size
    out = np.zeros((oh, ow, 3), np.float32)
    pad = (C.c_double * 4)()
    _check(lib().mi_image_to_tensor(device, C.c_void_p(image.ctypes.data), w, h, stride, C.byref(roi) if roi is not None else None,
                                    ow, oh, int(keep_aspect_ratio), output_range[0], output_range[1], int(flip_horizontal),
                                    C.c_void_p(out.ctypes.data), pad, MI_MEM_HOST, None))
    return out, tuple(pad)
