//! face_landmark.rs:168-306 of the reference.
use crate::types::{Detection, Image, Landmark, Rect};
use crate::{check, ffi, roi_ptr};
use anyhow::Error;
use std::ffi::CString;

/// `face_detection_to_roi(face_detection, image_size, size_mode)` — face_landmark.rs:180-198: rotation from the two eye
/// keypoints, scale 1.5, `SizeMode::SquareLong` (the only mode the reference's callers use; `None` selects it).
pub fn face_detection_to_roi(face_detection: Detection, image_size: (i32, i32), size_mode: Option<()>) -> Result<Rect, Error> {
    let _ = size_mode;
    let det = face_detection.to_mi();
    let mut out = ffi::mi_rect { x_center: 0.0, y_center: 0.0, width: 0.0, height: 0.0, rotation: 0.0, normalized: 1 };
    check(unsafe { ffi::mi_face_detection_to_roi(&det, image_size.0, image_size.1, &mut out) })?; // Err: "bbox must be normalized"
    Ok(Rect::from_mi(&out))
}

pub struct FaceLandmark {
    handle: *mut ffi::mi_fl,
}

unsafe impl Send for FaceLandmark {}
unsafe impl Sync for FaceLandmark {}

impl FaceLandmark {
    /// `FaceLandmark::new(model_path)` — face_landmark.rs:208-222.  `model_path` is the model FILE (default
    /// "./models/face_landmark.tflite"); fails when the mesh output is narrower than 1404 values (244-247).
    pub fn new(model_path: Option<String>) -> Result<FaceLandmark, Error> {
        Self::new_on_device(model_path, 0)
    }

    pub fn new_on_device(model_path: Option<String>, device: i32) -> Result<FaceLandmark, Error> {
        let path = match model_path {
            Some(p) => Some(CString::new(p)?),
            None => None,
        };
        let mut handle: *mut ffi::mi_fl = std::ptr::null_mut();
        check(unsafe { ffi::mi_fl_create(path.as_ref().map_or(std::ptr::null(), |p| p.as_ptr()), device, &mut handle) })?;
        Ok(FaceLandmark { handle })
    }

    /// `infer(&self, image, roi) -> Result<Vec<Landmark>>` — face_landmark.rs:232-306: rotated-ROI warp to 192x192 in [0, 1],
    /// network, face flag (`sigmoid(flag) <= 0.5` gives the empty `Vec`, 292-296), `project_landmarks` back to the picture.
    pub fn infer<'a, I>(&self, image: I, roi: Option<Rect>) -> Result<Vec<Landmark>, Error>
    where
        I: TryInto<Image<'a>>,
        I::Error: Into<Error>,
    {
        let image: Image<'a> = image.try_into().map_err(Into::into)?;
        let c_roi = roi.map(|r| r.to_mi());
        let mut out = vec![ffi::mi_landmark::default(); ffi::MI_NUM_FACE_LANDMARKS];
        let mut n: i32 = 0;
        check(unsafe {
            ffi::mi_fl_infer_image(self.handle, image.data.as_ptr(), image.width, image.height, image.stride, roi_ptr(&c_roi),
                                   out.as_mut_ptr(), out.len() as i32, &mut n)
        })?;
        Ok(out[..n.max(0) as usize].iter().map(|l| Landmark::new(l.x, l.y, l.z)).collect())
    }
}

impl Drop for FaceLandmark {
    fn drop(&mut self) {
        unsafe { ffi::mi_fl_free(self.handle) }
    }
}
