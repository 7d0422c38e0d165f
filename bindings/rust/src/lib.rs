//! Drop-in for okieraised/rs-face-detection-tflite's `FaceDetection::infer` / `FaceLandmark::infer` /
//! `IrisLandmark::infer` on an AMD MI355X: the same public names and signatures, every body a call through the C ABI of
//! `include/mi_face.h` into libmiface.so (hand-written HIP kernels for gfx950; no TFLite, no OpenCV).
//!
//! **Never compiled**: the image this repository is built in has no Rust toolchain.  The C ABI underneath is exercised end to
//! end by the Python (ctypes) and C++ (`include/mi_face.hpp`) mirrors in the same repository.
//!
//! Reference files mirrored (paths relative to the reference's `src/face_detection_lite/`):
//! `types.rs` -> [`types`], `face_detection.rs:117-267` -> [`face_detection`], `face_landmark.rs:168-306` ->
//! [`face_landmark`], `iris_landmark.rs:115-292,380-398` -> [`iris_landmark`], `utils.rs:8-21` -> [`utils`].
//!
//! Concurrency: the reference's handles are immutable after `new`, and `infer(&self)` may run on several threads.  The
//! handles here own device state (activation arena, replay graphs), so libmiface serialises the calls made on one handle
//! with an internal mutex; they are therefore `Send + Sync`, and one handle per worker thread is the way to overlap calls.
pub mod ffi;
pub mod types;

pub mod face_detection;
pub mod face_landmark;
pub mod iris_landmark;
pub mod utils;

pub use face_detection::{FaceDetection, FaceDetectionModel};
pub use face_landmark::{face_detection_to_roi, FaceLandmark};
pub use iris_landmark::{iris_roi_from_face_landmarks, update_face_landmarks_with_iris_results, IrisLandmark};
pub use types::{BBox, Detection, Image, IrisResults, Landmark, Rect};

use anyhow::Error;
use std::ffi::CStr;

/// `mi_last_error()` of the calling thread as an `anyhow::Error` (what the reference returns for I/O, model and shape
/// problems; its panics — SURVEY.md §5 — arrive here as errors too: nothing aborts across the ABI).
pub(crate) fn last_error(code: i32) -> Error {
    let msg = unsafe {
        let p = ffi::mi_last_error();
        if p.is_null() {
            String::from("unknown error")
        } else {
            CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    };
    Error::msg(format!("mi_face error {}: {}", code, msg))
}

pub(crate) fn check(code: i32) -> Result<(), Error> {
    if code == ffi::MI_OK {
        Ok(())
    } else {
        Err(last_error(code))
    }
}

/// `Option<Rect>` -> nullable `const mi_rect *`
pub(crate) fn roi_ptr(slot: &Option<ffi::mi_rect>) -> *const ffi::mi_rect {
    match slot {
        Some(r) => r as *const ffi::mi_rect,
        None => std::ptr::null(),
    }
}
