"""MI355X-native BlazeFace / face-mesh / iris inference path (drop-in for okieraised/rs-face-detection-tflite's
`FaceDetection::infer` / `FaceLandmark::infer` / `IrisLandmark::infer`).

Layout: csrc/ (HIP kernels for gfx950 + C++ host + the C ABI of include/mi_face.h, built into libmiface.so by build.sh)
and api.py (host-side mirror of the reference's public API over that C ABI).  The directory name carries a hyphen, so
import it as `rs_face_detection_tflite_amd` (shim module at the repo root).
"""
from .api import (  # noqa: F401
    DEFAULT_MODEL_DIR, EXPORTS, LIB_PATH, MI_MEM_DEVICE, MI_MEM_HOST, Detection, FaceDetection, FaceDetectionModel,
    FaceLandmark, IrisLandmark, IrisResults, Landmark, MiError, Model, PinnedBuffer, Pipeline, Rect, device_count, face_detection_to_roi,
    bbox_from_landmarks, bbox_to_roi, convert_image_to_mat, image_to_tensor, iris_roi_from_face_landmarks, jpeg_info, lib, plan_describe, dist_broadcast_bytes, streams_create_distinct, streams_destroy, update_face_landmarks_with_iris_results,
)
