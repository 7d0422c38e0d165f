//! iris_landmark.rs:115-292, 380-398 of the reference.
use crate::types::{Image, IrisResults, Landmark, Rect};
use crate::{check, ffi, roi_ptr};
use anyhow::Error;
use std::ffi::CString;

/// `iris_roi_from_face_landmarks(face_landmarks, image_size) -> (left_eye_roi, right_eye_roi)` — iris_landmark.rs:268-292
/// (landmarks 33/133 and 362/263, scale 2.3, `SquareLong`).
pub fn iris_roi_from_face_landmarks(face_landmarks: Vec<Landmark>, image_size: (i32, i32)) -> Result<(Rect, Rect), Error> {
    if face_landmarks.len() < ffi::MI_NUM_FACE_LANDMARKS {
        return Err(Error::msg("expected 468 face landmarks")); // the reference would panic indexing [362]
    }
    let lm: Vec<ffi::mi_landmark> = face_landmarks.iter().map(|l| l.to_mi()).collect();
    let blank = ffi::mi_rect { x_center: 0.0, y_center: 0.0, width: 0.0, height: 0.0, rotation: 0.0, normalized: 1 };
    let (mut left, mut right) = (blank, blank);
    check(unsafe { ffi::mi_iris_roi_from_face_landmarks(lm.as_ptr(), image_size.0, image_size.1, &mut left, &mut right) })?;
    Ok((Rect::from_mi(&left), Rect::from_mi(&right)))
}

/// `update_face_landmarks_with_iris_results` — iris_landmark.rs:380-398: the 71 eye-contour points of each eye replace the
/// mesh points they refine (index maps 64-95), left eye first.
pub fn update_face_landmarks_with_iris_results(face_landmarks: Vec<Landmark>, iris_data_left: IrisResults, iris_data_right: IrisResults)
                                               -> Result<Vec<Landmark>, Error> {
    if face_landmarks.len() != ffi::MI_NUM_FACE_LANDMARKS {
        return Err(Error::msg("unexpected number of items in face_landmarks"));
    }
    if iris_data_left.contour.len() != ffi::MI_NUM_EYE_LANDMARKS || iris_data_right.contour.len() != ffi::MI_NUM_EYE_LANDMARKS {
        return Err(Error::msg("expected 71 contour landmarks per eye"));
    }
    let face: Vec<ffi::mi_landmark> = face_landmarks.iter().map(|l| l.to_mi()).collect();
    let left: Vec<ffi::mi_landmark> = iris_data_left.contour.iter().map(|l| l.to_mi()).collect();
    let right: Vec<ffi::mi_landmark> = iris_data_right.contour.iter().map(|l| l.to_mi()).collect();
    let mut out = vec![ffi::mi_landmark::default(); ffi::MI_NUM_FACE_LANDMARKS];
    check(unsafe { ffi::mi_update_face_landmarks_with_iris_results(face.as_ptr(), left.as_ptr(), right.as_ptr(), out.as_mut_ptr()) })?;
    Ok(out.iter().map(|l| Landmark::new(l.x, l.y, l.z)).collect())
}

pub struct IrisLandmark {
    handle: *mut ffi::mi_iris,
}

unsafe impl Send for IrisLandmark {}
unsafe impl Sync for IrisLandmark {}

impl IrisLandmark {
    /// `IrisLandmark::new(model_path)` — iris_landmark.rs:142-156 (model FILE, default "./models/iris_landmark.tflite");
    /// fails unless the outputs are 213 and 15 values wide (172-184).
    pub fn new(model_path: Option<String>) -> Result<IrisLandmark, Error> {
        Self::new_on_device(model_path, 0)
    }

    pub fn new_on_device(model_path: Option<String>, device: i32) -> Result<IrisLandmark, Error> {
        let path = match model_path {
            Some(p) => Some(CString::new(p)?),
            None => None,
        };
        let mut handle: *mut ffi::mi_iris = std::ptr::null_mut();
        check(unsafe { ffi::mi_iris_create(path.as_ref().map_or(std::ptr::null(), |p| p.as_ptr()), device, &mut handle) })?;
        Ok(IrisLandmark { handle })
    }

    /// `infer(&self, image, roi, is_right_eye) -> Result<IrisResults>` — iris_landmark.rs:158-248: letterboxed ROI warp to
    /// 64x64 in [0, 1] (flipped for the right eye), network, both outputs projected back (x mirrored for the right eye).
    pub fn infer<'a, I>(&self, image: I, roi: Option<Rect>, is_right_eye: Option<bool>) -> Result<IrisResults, Error>
    where
        I: TryInto<Image<'a>>,
        I::Error: Into<Error>,
    {
        let image: Image<'a> = image.try_into().map_err(Into::into)?;
        let c_roi = roi.map(|r| r.to_mi());
        let mut contour = vec![ffi::mi_landmark::default(); ffi::MI_NUM_EYE_LANDMARKS];
        let mut iris = vec![ffi::mi_landmark::default(); ffi::MI_NUM_IRIS_LANDMARKS];
        check(unsafe {
            ffi::mi_iris_infer_image(self.handle, image.data.as_ptr(), image.width, image.height, image.stride, roi_ptr(&c_roi),
                                     is_right_eye.unwrap_or(false) as i32, contour.as_mut_ptr(), iris.as_mut_ptr())
        })?;
        Ok(IrisResults::new(contour.iter().map(|l| Landmark::new(l.x, l.y, l.z)).collect(),
                            iris.iter().map(|l| Landmark::new(l.x, l.y, l.z)).collect()))
    }
}

impl IrisLandmark {
    /// `infer` over a batch (`mi_iris_infer_images`): item `i` reads frame `i / items_per_frame` with eye ROI `rois[i]` and flip
    /// `is_right_eye[i]` (`is_right_eye` empty: no flips; `rois` empty: whole frames, `items_per_frame` 1) — iris_landmark.rs:158-248
    /// per item, the warp to 64x64 on the device.
    pub fn infer_batch(&self, frames: &[u8], batch: usize, width: i32, height: i32, stride: i32, rois: &[Rect], is_right_eye: &[bool],
                       items_per_frame: usize) -> Result<Vec<IrisResults>, Error> {
        if batch == 0 || items_per_frame == 0 || width <= 0 || height <= 0 || stride <= 0 || (stride as i64) < 3 * width as i64 {
            return Err(Error::msg("frames must hold batch frames of height rows of stride bytes"));
        }
        let n = batch.checked_mul(items_per_frame).filter(|n| *n <= i32::MAX as usize).ok_or_else(|| Error::msg("too many items"))?;
        if (rois.is_empty() && items_per_frame != 1) || (!rois.is_empty() && rois.len() != n) || (!is_right_eye.is_empty() && is_right_eye.len() != n) {
            return Err(Error::msg("rois / is_right_eye must hold one entry per item"));
        }
        let (w, h, s) = (width as usize, height as usize, stride as usize);
        let need = s.checked_mul(h).and_then(|f| f.checked_mul(batch - 1)).and_then(|x| x.checked_add(s * (h - 1))).and_then(|x| x.checked_add(3 * w));
        if need.map_or(true, |x| frames.len() < x) {
            return Err(Error::msg("frames must hold batch frames of height rows of stride bytes"));
        }
        let c_rois: Vec<ffi::mi_rect> = rois.iter().map(|r| r.to_mi()).collect();
        let flips: Vec<i32> = is_right_eye.iter().map(|b| *b as i32).collect();
        let mut contour = vec![0f32; n * 3 * ffi::MI_NUM_EYE_LANDMARKS];
        let mut iris = vec![0f32; n * 3 * ffi::MI_NUM_IRIS_LANDMARKS];
        check(unsafe {
            ffi::mi_iris_infer_images(self.handle, frames.as_ptr(), batch as i32, width, height, stride,
                                      if c_rois.is_empty() { std::ptr::null() } else { c_rois.as_ptr() },
                                      if flips.is_empty() { std::ptr::null() } else { flips.as_ptr() }, items_per_frame as i32,
                                      contour.as_mut_ptr(), iris.as_mut_ptr(), ffi::MI_MEM_HOST, std::ptr::null_mut())
        })?;
        let lms = |v: &[f32]| v.chunks_exact(3).map(|p| Landmark::new(p[0] as f64, p[1] as f64, p[2] as f64)).collect::<Vec<_>>();
        Ok((0..n)
            .map(|i| {
                IrisResults::new(lms(&contour[i * 3 * ffi::MI_NUM_EYE_LANDMARKS..(i + 1) * 3 * ffi::MI_NUM_EYE_LANDMARKS]),
                                 lms(&iris[i * 3 * ffi::MI_NUM_IRIS_LANDMARKS..(i + 1) * 3 * ffi::MI_NUM_IRIS_LANDMARKS]))
            })
            .collect())
    }
}

impl Drop for IrisLandmark {
    fn drop(&mut self) {
        unsafe { ffi::mi_iris_free(self.handle) }
    }
}
