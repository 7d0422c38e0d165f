//! Raw declarations of `include/mi_face.h` (the drop-in boundary).  One `extern "C"` item per C entry point the three
//! `infer` paths and their helpers use; layouts are `#[repr(C)]` mirrors of `mi_detection`, `mi_rect`, `mi_landmark`.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_float, c_int, c_void};

pub const MI_OK: c_int = 0;
pub const MI_MEM_HOST: c_int = 0;
pub const MI_MEM_DEVICE: c_int = 1;
pub const MI_NUM_FACE_LANDMARKS: usize = 468;
pub const MI_NUM_EYE_LANDMARKS: usize = 71;
pub const MI_NUM_IRIS_LANDMARKS: usize = 5;

/// `Detection { data: Array2<f32>[8,2], score: f32 }` — types.rs:189-193
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct mi_detection {
    pub data: [c_float; 16],
    pub score: c_float,
}

/// `Rect` — types.rs:24-36
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct mi_rect {
    pub x_center: c_double,
    pub y_center: c_double,
    pub width: c_double,
    pub height: c_double,
    pub rotation: c_double,
    pub normalized: c_int,
}

/// `Landmark { x, y, z: f64 }` — types.rs:176-187
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct mi_landmark {
    pub x: c_double,
    pub y: c_double,
    pub z: c_double,
}

#[repr(C)] pub struct mi_fd { _private: [u8; 0] }
#[repr(C)] pub struct mi_fl { _private: [u8; 0] }
#[repr(C)] pub struct mi_iris { _private: [u8; 0] }
#[repr(C)] pub struct mi_pipeline { _private: [u8; 0] }

extern "C" {
    pub fn mi_last_error() -> *const c_char;
    pub fn mi_device_count() -> c_int;

    // FaceDetection — face_detection.rs:146-267
    pub fn mi_fd_create(kind: c_int, model_dir: *const c_char, device: c_int, out: *mut *mut mi_fd) -> c_int;
    pub fn mi_fd_create_from_bytes(kind: c_int, tflite: *const u8, nbytes: usize, device: c_int, out: *mut *mut mi_fd) -> c_int;
    pub fn mi_fd_free(h: *mut mi_fd);
    pub fn mi_fd_input_size(h: *const mi_fd, width: *mut c_int, height: *mut c_int) -> c_int;
    pub fn mi_fd_infer_image(h: *mut mi_fd, rgb: *const u8, width: c_int, height: c_int, stride: c_int, roi: *const mi_rect,
                             out: *mut mi_detection, cap: c_int, count: *mut c_int) -> c_int;
    pub fn mi_fd_infer_tensor(h: *mut mi_fd, input: *const c_float, batch: c_int, padding: *const c_double, out: *mut mi_detection,
                              cap_per_frame: c_int, counts: *mut c_int, mem: c_int, stream: *mut c_void) -> c_int;
    pub fn mi_fd_infer_images(h: *mut mi_fd, frames: *const u8, batch: c_int, width: c_int, height: c_int, stride: c_int,
                              rois: *const mi_rect, out: *mut mi_detection, cap_per_frame: c_int, counts: *mut c_int, mem: c_int,
                              stream: *mut c_void) -> c_int;
    pub fn mi_fd_submit_images(h: *mut mi_fd, slot: c_int, frames: *const u8, batch: c_int, width: c_int, height: c_int, stride: c_int,
                               cap_per_frame: c_int) -> c_int;
    pub fn mi_fd_collect(h: *mut mi_fd, slot: c_int, out: *mut mi_detection, counts: *mut c_int) -> c_int;
    pub fn mi_fd_submit_jpeg(h: *mut mi_fd, slot: c_int, bytes: *const u8, nbytes: usize, cap: c_int) -> c_int;
    pub fn mi_fd_collect_jpeg(h: *mut mi_fd, slot: c_int, out: *mut mi_detection, cap: c_int, count: *mut c_int, width: *mut c_int,
                              height: *mut c_int) -> c_int;
    pub fn mi_host_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn mi_host_free(p: *mut c_void);

    // FaceLandmark — face_landmark.rs:200-306
    pub fn mi_fl_create(model_path: *const c_char, device: c_int, out: *mut *mut mi_fl) -> c_int;
    pub fn mi_fl_free(h: *mut mi_fl);
    pub fn mi_fl_infer_image(h: *mut mi_fl, rgb: *const u8, width: c_int, height: c_int, stride: c_int, roi: *const mi_rect,
                             out: *mut mi_landmark, cap: c_int, count: *mut c_int) -> c_int;
    pub fn mi_fl_infer_tensor(h: *mut mi_fl, input: *const c_float, batch: c_int, rois: *const mi_rect, image_sizes: *const c_int,
                              landmarks: *mut c_float, present: *mut c_int, raw_flags: *mut c_float, mem: c_int, stream: *mut c_void) -> c_int;
    pub fn mi_fl_infer_images(h: *mut mi_fl, frames: *const u8, batch: c_int, width: c_int, height: c_int, stride: c_int, rois: *const mi_rect,
                              items_per_frame: c_int, landmarks: *mut c_float, present: *mut c_int, raw_flags: *mut c_float, mem: c_int,
                              stream: *mut c_void) -> c_int;

    pub fn mi_fl_submit_images(h: *mut mi_fl, slot: c_int, frames: *const u8, batch: c_int, width: c_int, height: c_int, stride: c_int,
                               rois: *const mi_rect, items_per_frame: c_int) -> c_int;
    pub fn mi_fl_collect(h: *mut mi_fl, slot: c_int, landmarks: *mut c_float, present: *mut c_int, raw_flags: *mut c_float) -> c_int;

    // IrisLandmark — iris_landmark.rs:130-248
    pub fn mi_iris_create(model_path: *const c_char, device: c_int, out: *mut *mut mi_iris) -> c_int;
    pub fn mi_iris_free(h: *mut mi_iris);
    pub fn mi_iris_infer_image(h: *mut mi_iris, rgb: *const u8, width: c_int, height: c_int, stride: c_int, roi: *const mi_rect,
                               is_right_eye: c_int, contour71: *mut mi_landmark, iris5: *mut mi_landmark) -> c_int;
    pub fn mi_iris_infer_images(h: *mut mi_iris, frames: *const u8, batch: c_int, width: c_int, height: c_int, stride: c_int, rois: *const mi_rect,
                                is_right_eye: *const c_int, items_per_frame: c_int, contour: *mut c_float, iris: *mut c_float, mem: c_int,
                                stream: *mut c_void) -> c_int;

    // helpers the reference exports next to the three structs
    pub fn mi_face_detection_to_roi(det: *const mi_detection, image_w: c_int, image_h: c_int, out: *mut mi_rect) -> c_int;
    pub fn mi_iris_roi_from_face_landmarks(landmarks468: *const mi_landmark, image_w: c_int, image_h: c_int, left_eye: *mut mi_rect,
                                           right_eye: *mut mi_rect) -> c_int;
    pub fn mi_update_face_landmarks_with_iris_results(face468: *const mi_landmark, left71: *const mi_landmark, right71: *const mi_landmark,
                                                      out468: *mut mi_landmark) -> c_int;
    pub fn mi_bbox_to_roi(bbox: *const c_double, image_w: c_int, image_h: c_int, rotation_keypoints: *const c_double, scale_x: c_double,
                          scale_y: c_double, size_mode: c_int, out: *mut mi_rect) -> c_int;
    pub fn mi_bbox_from_landmarks(landmarks: *const mi_landmark, count: c_int, bbox_out: *mut c_double) -> c_int;
    pub fn mi_jpeg_info(bytes: *const u8, nbytes: usize, width: *mut c_int, height: *mut c_int) -> c_int;
    pub fn mi_jpeg_decode_rgb(device: c_int, bytes: *const u8, nbytes: usize, rgb: *mut u8, cap_bytes: usize, width: *mut c_int,
                              height: *mut c_int, mem: c_int, stream: *mut c_void) -> c_int;

    // multi-GPU: the frozen .tflite bytes from `root` to every rank over RCCL (one process per GPU)
    pub fn mi_dist_broadcast_bytes(id_path: *const c_char, rank: c_int, world: c_int, root: c_int, device: c_int, buf: *mut u8, nbytes: usize,
                                   timeout_ms: c_int) -> c_int;

    // two batches in flight: n hipStream_t's tested to sit on distinct hardware queues
    pub fn mi_streams_create_distinct(device: c_int, n: c_int, streams: *mut *mut c_void) -> c_int;
    pub fn mi_streams_destroy(device: c_int, n: c_int, streams: *mut *mut c_void) -> c_int;

    // batched detector -> mesh -> iris flow on the device (no counterpart in the reference: lib.rs:24-40 per frame)
    pub fn mi_pipeline_create(fd_kind: c_int, model_dir: *const c_char, device: c_int, out: *mut *mut mi_pipeline) -> c_int;
    pub fn mi_pipeline_free(p: *mut mi_pipeline);
    pub fn mi_pipeline_run(p: *mut mi_pipeline, frames: *const u8, batch: c_int, width: c_int, height: c_int, stride: c_int,
                           faces: *mut mi_detection, face_counts: *mut c_int, landmarks: *mut c_float, present: *mut c_int,
                           eyes: *mut c_float, mem: c_int, stream: *mut c_void) -> c_int;
}
