//! face_detection.rs:117-267 of the reference.
use crate::types::{Detection, Image, Rect};
use crate::{check, ffi, roi_ptr};
use anyhow::Error;
use std::ffi::CString;

/// face_detection.rs:117-123 (discriminants = `MI_FD_*` of mi_face.h)
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
#[repr(i32)]
pub enum FaceDetectionModel {
    FrontCamera = 0,
    BackCamera = 1,
    Short = 2,
    Full = 3,
    FullSparse = 4,
}

pub struct FaceDetection {
    handle: *mut ffi::mi_fd,
}

// libmiface serialises calls on one handle (mi_face.h, conventions)
unsafe impl Send for FaceDetection {}
unsafe impl Sync for FaceDetection {}

impl FaceDetection {
    /// `FaceDetection::new(model_type, model_path)` — face_detection.rs:153-195.  `model_path` is the model DIRECTORY
    /// (default "./models"); the file name follows from `model_type` (125-129, 163-185).
    pub fn new(model_type: FaceDetectionModel, model_path: Option<String>) -> Result<FaceDetection, Error> {
        Self::new_on_device(model_type, model_path, 0)
    }

    /// Same, bound to the HIP device `device` (one process per GPU in the sharded configurations).
    pub fn new_on_device(model_type: FaceDetectionModel, model_path: Option<String>, device: i32) -> Result<FaceDetection, Error> {
        let dir = CString::new(model_path.unwrap_or_else(|| String::from("./models")))?;
        let mut handle: *mut ffi::mi_fd = std::ptr::null_mut();
        check(unsafe { ffi::mi_fd_create(model_type as i32, dir.as_ptr(), device, &mut handle) })?;
        Ok(FaceDetection { handle })
    }

    /// The handle from the model's BYTES instead of a file (`mi_fd_create_from_bytes`): what a rank that received the frozen
    /// `.tflite` over RCCL (`mi_dist_broadcast_bytes`) builds its detector from — the counterpart of
    /// `FlatBufferModel::build_from_file` (face_detection.rs:188) on ranks that hold no file.
    pub fn from_bytes(model_type: FaceDetectionModel, tflite: &[u8], device: i32) -> Result<FaceDetection, Error> {
        let mut handle: *mut ffi::mi_fd = std::ptr::null_mut();
        check(unsafe { ffi::mi_fd_create_from_bytes(model_type as i32, tflite.as_ptr(), tflite.len(), device, &mut handle) })?;
        Ok(FaceDetection { handle })
    }

    /// `infer(&self, image: &Mat, roi) -> Result<Vec<Detection>>` — face_detection.rs:205-267.  `image` is anything that turns
    /// into an `Image` view: `&Mat` (feature `opencv`), `&RgbImage`, `&Image` or an `Image` — so the reference's call
    /// `face_detection.infer(&mat, None)` compiles unchanged.  Flow: image_to_tensor (letterbox to the
    /// model size, [-1, 1]) -> network -> decode_boxes -> sigmoid -> threshold -> weighted NMS -> letterbox removal, all on
    /// the GPU.  Detections come back in descending head-score order, normalised to the picture.
    pub fn infer<'a, I>(&self, image: I, roi: Option<Rect>) -> Result<Vec<Detection>, Error>
    where
        I: TryInto<Image<'a>>,
        I::Error: Into<Error>,
    {
        let image: Image<'a> = image.try_into().map_err(Into::into)?;
        let c_roi = roi.map(|r| r.to_mi());
        let mut cap = 64usize;
        loop {
            let mut out = vec![ffi::mi_detection { data: [0.0; 16], score: 0.0 }; cap];
            let mut n: i32 = 0;
            check(unsafe {
                ffi::mi_fd_infer_image(self.handle, image.data.as_ptr(), image.width, image.height, image.stride, roi_ptr(&c_roi),
                                       out.as_mut_ptr(), cap as i32, &mut n)
            })?; // MI_ERANGE = the letterbox assert! of transform.rs:121-122
            let n = n.max(0) as usize;
            if n <= cap {
                return Ok(out[..n].iter().map(Detection::from_mi).collect());
            }
            cap = n; // the reference returns every detection: ask again with room for all of them
        }
    }

    /// Batched form for tensors already at the model's resolution (BASELINE configs 2 and 4): `input` is f32 NHWC
    /// `[batch, H, W, 3]` in [-1, 1] in HOST memory; `padding` = one `(left, top, right, bottom)` per frame or empty.
    pub fn infer_tensor(&self, input: &[f32], batch: usize, padding: &[[f64; 4]], cap_per_frame: usize) -> Result<Vec<Vec<Detection>>, Error> {
        if !padding.is_empty() && padding.len() != batch {
            return Err(Error::msg("padding must be empty or hold one entry per frame"));
        }
        // the C side reads batch * H * W * 3 floats: a shorter slice must never reach it from safe code
        let (mut w, mut h) = (0i32, 0i32);
        check(unsafe { ffi::mi_fd_input_size(self.handle, &mut w, &mut h) })?;
        let need = batch.checked_mul(w as usize * h as usize * 3).ok_or_else(|| Error::msg("batch too large"))?;
        if batch == 0 || batch > i32::MAX as usize || cap_per_frame == 0 || cap_per_frame > i32::MAX as usize || input.len() != need {
            return Err(Error::msg(format!("input must hold batch x {} x {} x 3 floats ({}), got {}", h, w, need, input.len())));
        }
        let mut out = vec![ffi::mi_detection { data: [0.0; 16], score: 0.0 }; batch * cap_per_frame];
        let mut counts = vec![0i32; batch];
        let pad_ptr = if padding.is_empty() { std::ptr::null() } else { padding.as_ptr() as *const f64 };
        check(unsafe {
            ffi::mi_fd_infer_tensor(self.handle, input.as_ptr(), batch as i32, pad_ptr, out.as_mut_ptr(), cap_per_frame as i32,
                                    counts.as_mut_ptr(), ffi::MI_MEM_HOST, std::ptr::null_mut())
        })?;
        Ok((0..batch)
            .map(|b| {
                let n = (counts[b].max(0) as usize).min(cap_per_frame);
                out[b * cap_per_frame..b * cap_per_frame + n].iter().map(Detection::from_mi).collect()
            })
            .collect())
    }
}

impl FaceDetection {
    /// `infer` over a batch of equally sized frames in ONE call (`mi_fd_infer_images`): u8 RGB frames cross the bus (a quarter of
    /// the f32 tensors), image_to_tensor runs on the device.  `frames` = `batch` frames of `height` rows of `stride` bytes.
    pub fn infer_batch(&self, frames: &[u8], batch: usize, width: i32, height: i32, stride: i32, cap_per_frame: usize) -> Result<Vec<Vec<Detection>>, Error> {
        // (sizes are validated before anything is computed from them: height 0 would underflow, the products may overflow)
        if batch == 0 || batch > i32::MAX as usize || cap_per_frame == 0 || cap_per_frame > i32::MAX as usize || width <= 0 || height <= 0 || stride <= 0
            || (stride as i64) < 3 * width as i64
        {
            return Err(Error::msg("frames must hold batch frames of height rows of stride bytes"));
        }
        let (w, h, s) = (width as usize, height as usize, stride as usize);
        let need = s.checked_mul(h).and_then(|f| f.checked_mul(batch - 1)).and_then(|x| x.checked_add(s * (h - 1))).and_then(|x| x.checked_add(3 * w));
        let slots = batch.checked_mul(cap_per_frame);
        let (need, slots) = match (need, slots) {
            (Some(n), Some(c)) if frames.len() >= n => (n, c),
            _ => return Err(Error::msg("frames must hold batch frames of height rows of stride bytes")),
        };
        let _ = need;
        let mut out = vec![ffi::mi_detection { data: [0.0; 16], score: 0.0 }; slots];
        let mut counts = vec![0i32; batch];
        check(unsafe {
            ffi::mi_fd_infer_images(self.handle, frames.as_ptr(), batch as i32, width, height, stride, std::ptr::null(), out.as_mut_ptr(),
                                    cap_per_frame as i32, counts.as_mut_ptr(), ffi::MI_MEM_HOST, std::ptr::null_mut())
        })?;
        Ok((0..batch)
            .map(|b| {
                let n = (counts[b].max(0) as usize).min(cap_per_frame);
                out[b * cap_per_frame..b * cap_per_frame + n].iter().map(Detection::from_mi).collect()
            })
            .collect())
    }
}

impl FaceDetection {
    /// `convert_image_to_mat` + `infer(&mat, None)` for a STREAM of encoded pictures (utils.rs:8-21 then face_detection.rs:205-267 — the first
    /// lines of the reference's own test, lib.rs:20-24), in two halves and two slots: `submit_jpeg` decodes the entropy-coded data on the
    /// calling thread while the GPU still works on the picture in the other slot and queues everything else (`mi_fd_submit_jpeg`);
    /// `collect_jpeg` waits for that slot and returns its detections and the picture's size.  216 us per picture sustained on an MI355X
    /// (one call at a time through `convert_image_to_mat` + `infer`: 425 us).
    pub fn submit_jpeg(&self, slot: i32, im_bytes: &[u8], cap: usize) -> Result<(), Error> {
        if cap == 0 || cap > i32::MAX as usize {
            return Err(Error::msg("cap must be positive"));
        }
        check(unsafe { ffi::mi_fd_submit_jpeg(self.handle, slot, im_bytes.as_ptr(), im_bytes.len(), cap as i32) })
    }

    /// The detections of the picture submitted to `slot` (at most `cap`, the value given to `submit_jpeg`) and its (width, height).
    pub fn collect_jpeg(&self, slot: i32, cap: usize) -> Result<(Vec<Detection>, (i32, i32)), Error> {
        if cap == 0 || cap > i32::MAX as usize {
            return Err(Error::msg("cap must be positive"));
        }
        let mut out = vec![ffi::mi_detection { data: [0.0; 16], score: 0.0 }; cap];
        let (mut n, mut w, mut h) = (0i32, 0i32, 0i32);
        check(unsafe { ffi::mi_fd_collect_jpeg(self.handle, slot, out.as_mut_ptr(), cap as i32, &mut n, &mut w, &mut h) })?;
        let n = (n.max(0) as usize).min(cap);
        Ok((out[..n].iter().map(Detection::from_mi).collect(), (w, h)))
    }
}

impl Drop for FaceDetection {
    fn drop(&mut self) {
        unsafe { ffi::mi_fd_free(self.handle) }
    }
}
