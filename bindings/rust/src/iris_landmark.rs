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

impl Drop for IrisLandmark {
    fn drop(&mut self) {
        unsafe { ffi::mi_iris_free(self.handle) }
    }
}
