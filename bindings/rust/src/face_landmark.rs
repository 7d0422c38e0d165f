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

impl FaceLandmark {
    /// `infer` over a batch (`mi_fl_infer_images`): `frames` holds `batch` 8UC3 RGB frames of `height` rows of `stride` bytes; item `i`
    /// reads frame `i / items_per_frame` with ROI `rois[i]` (`rois` empty: whole frames, `items_per_frame` 1).  One `Vec` per item,
    /// empty where the face flag fails.  The warp to 192x192 runs on the device: only the frames and the ROIs cross the bus.
    pub fn infer_batch(&self, frames: &[u8], batch: usize, width: i32, height: i32, stride: i32, rois: &[Rect], items_per_frame: usize)
        -> Result<Vec<Vec<Landmark>>, Error> {
        if batch == 0 || items_per_frame == 0 || width <= 0 || height <= 0 || stride <= 0 || (stride as i64) < 3 * width as i64 {
            return Err(Error::msg("frames must hold batch frames of height rows of stride bytes"));
        }
        let n = batch.checked_mul(items_per_frame).filter(|n| *n <= i32::MAX as usize).ok_or_else(|| Error::msg("too many items"))?;
        if (rois.is_empty() && items_per_frame != 1) || (!rois.is_empty() && rois.len() != n) {
            return Err(Error::msg("rois must hold one entry per item"));
        }
        let (w, h, s) = (width as usize, height as usize, stride as usize);
        let need = s.checked_mul(h).and_then(|f| f.checked_mul(batch - 1)).and_then(|x| x.checked_add(s * (h - 1))).and_then(|x| x.checked_add(3 * w));
        if need.map_or(true, |x| frames.len() < x) {
            return Err(Error::msg("frames must hold batch frames of height rows of stride bytes"));
        }
        let c_rois: Vec<ffi::mi_rect> = rois.iter().map(|r| r.to_mi()).collect();
        let mut lm = vec![0f32; n * 3 * ffi::MI_NUM_FACE_LANDMARKS];
        let mut present = vec![0i32; n];
        check(unsafe {
            ffi::mi_fl_infer_images(self.handle, frames.as_ptr(), batch as i32, width, height, stride,
                                    if c_rois.is_empty() { std::ptr::null() } else { c_rois.as_ptr() }, items_per_frame as i32, lm.as_mut_ptr(),
                                    present.as_mut_ptr(), std::ptr::null_mut(), ffi::MI_MEM_HOST, std::ptr::null_mut())
        })?;
        Ok((0..n)
            .map(|i| {
                if present[i] == 0 {
                    return Vec::new();
                }
                lm[i * 3 * ffi::MI_NUM_FACE_LANDMARKS..(i + 1) * 3 * ffi::MI_NUM_FACE_LANDMARKS]
                    .chunks_exact(3)
                    .map(|v| Landmark::new(v[0] as f64, v[1] as f64, v[2] as f64))
                    .collect()
            })
            .collect())
    }
}

impl Drop for FaceLandmark {
    fn drop(&mut self) {
        unsafe { ffi::mi_fl_free(self.handle) }
    }
}
