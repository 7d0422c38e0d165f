//! utils.rs:8-21 of the reference: `convert_image_to_mat(im_bytes)` for JPEG input — Huffman decoding on the host, IDCT /
//! chroma upsampling / colour conversion on the GPU, bit-identical to libjpeg-turbo (what `cv::imdecode` runs).
use crate::{check, ffi};
use anyhow::Error;

/// Owning 8UC3 RGB picture; `image()` borrows it for the `infer` calls.
pub struct RgbImage {
    pub data: Vec<u8>,
    pub width: i32,
    pub height: i32,
}

impl RgbImage {
    pub fn image(&self) -> crate::types::Image<'_> {
        crate::types::Image { data: &self.data, width: self.width, height: self.height, stride: 3 * self.width }
    }
}

/// Baseline / extended-sequential Huffman JPEG, 8 bit, grey or YCbCr (h1v1, h2v1, h2v2).  Anything else (progressive,
/// arithmetic coding, other containers) is an error: keep `imdecode` for those.
pub fn convert_image_to_mat(im_bytes: &[u8]) -> Result<RgbImage, Error> {
    let (mut w, mut h) = (0i32, 0i32);
    check(unsafe { ffi::mi_jpeg_info(im_bytes.as_ptr(), im_bytes.len(), &mut w, &mut h) })?;
    let mut data = vec![0u8; 3 * w as usize * h as usize];
    check(unsafe {
        ffi::mi_jpeg_decode_rgb(0, im_bytes.as_ptr(), im_bytes.len(), data.as_mut_ptr(), data.len(), &mut w, &mut h, ffi::MI_MEM_HOST,
                                std::ptr::null_mut())
    })?;
    Ok(RgbImage { data, width: w, height: h })
}
