//! utils.rs:8-21 of the reference: `convert_image_to_mat(im_bytes)` for JPEG input — Huffman decoding on the host, IDCT /
//! chroma upsampling / colour conversion on the GPU, bit-identical to libjpeg-turbo (what `cv::imdecode` runs).
use crate::{check, ffi};
use anyhow::Error;

/// Owning 8UC3 RGB picture (what the reference holds as a `Mat`); `infer(&picture, ..)` borrows it through
/// `TryFrom<&RgbImage> for Image`.  Private fields: the buffer always matches the geometry.
pub struct RgbImage {
    data: Vec<u8>,
    width: i32,
    height: i32,
}

impl RgbImage {
    pub fn new(data: Vec<u8>, width: i32, height: i32) -> Result<RgbImage, Error> {
        if width <= 0 || height <= 0 || data.len() != 3 * width as usize * height as usize {
            return Err(Error::msg("RGB buffer does not match width x height x 3"));
        }
        Ok(RgbImage { data, width, height })
    }
    pub fn data(&self) -> &[u8] {
        &self.data
    }
    pub fn width(&self) -> i32 {
        self.width
    }
    pub fn height(&self) -> i32 {
        self.height
    }
    pub fn image(&self) -> crate::types::Image<'_> {
        crate::types::Image { data: &self.data, width: self.width, height: self.height, stride: 3 * self.width }
    }
}

/// Baseline / extended-sequential / progressive Huffman JPEG, 8 bit, grey or YCbCr (h1v1, h2v1, h2v2).  Anything else (
/// arithmetic coding, other containers) is an error: keep `imdecode` for those ("Pictures libmiface does not
/// decode": fall back to the reference's own `convert_image_to_mat` and pass the `Mat` to `infer` under `--features opencv`).
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
