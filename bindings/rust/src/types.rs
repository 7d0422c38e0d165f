//! Value types crossing the API — same names, fields and methods as the reference's types.rs:24-246 and
//! iris_landmark.rs:115-129.  `Detection::data` is `[[f32; 2]; 8]` here where the reference holds an `Array2<f32>` of
//! shape (8, 2): row 0 = (xmin, ymin), row 1 = (xmax, ymax), rows 2..8 = the six keypoints in `FaceIndex` order.
use crate::ffi;

/// types.rs:24-36
#[derive(Debug, Clone, Copy)]
pub struct Rect {
    pub x_center: f64,
    pub y_center: f64,
    pub width: f64,
    pub height: f64,
    /// rotation in radians (clockwise)
    pub rotation: f64,
    /// properties are relative to the image size
    pub normalized: bool,
}

impl Rect {
    pub fn new(x_center: f64, y_center: f64, width: f64, height: f64, rotation: f64, normalized: bool) -> Self {
        Self { x_center, y_center, width, height, rotation, normalized }
    }

    /// types.rs:52-59
    pub fn size(&self) -> (f64, f64) {
        if self.normalized {
            (self.width, self.height)
        } else {
            (self.width as i32 as f64, self.height as i32 as f64)
        }
    }

    /// types.rs:62-77
    pub fn scaled(&self, size: (f64, f64), normalize: bool) -> Rect {
        if self.normalized == normalize {
            return *self;
        }
        let (sx, sy) = if normalize { (1.0 / size.0, 1.0 / size.1) } else { size };
        Rect {
            x_center: self.x_center * sx,
            y_center: self.y_center * sy,
            width: self.width * sx,
            height: self.height * sy,
            rotation: self.rotation,
            normalized: normalize,
        }
    }

    pub(crate) fn to_mi(self) -> ffi::mi_rect {
        ffi::mi_rect {
            x_center: self.x_center,
            y_center: self.y_center,
            width: self.width,
            height: self.height,
            rotation: self.rotation,
            normalized: self.normalized as i32,
        }
    }

    pub(crate) fn from_mi(r: &ffi::mi_rect) -> Rect {
        Rect::new(r.x_center, r.y_center, r.width, r.height, r.rotation, r.normalized != 0)
    }
}

/// types.rs:99-174 (the parts callers of `infer` use)
#[derive(Debug, Clone, Copy)]
pub struct BBox {
    pub xmin: f64,
    pub ymin: f64,
    pub xmax: f64,
    pub ymax: f64,
}

impl BBox {
    pub fn new(xmin: f64, ymin: f64, xmax: f64, ymax: f64) -> Self {
        Self { xmin, ymin, xmax, ymax }
    }
    pub fn as_tuple(&self) -> (f64, f64, f64, f64) {
        (self.xmin, self.ymin, self.xmax, self.ymax)
    }
    pub fn width(&self) -> f64 {
        self.xmax - self.xmin
    }
    pub fn height(&self) -> f64 {
        self.ymax - self.ymin
    }
    pub fn empty(&self) -> bool {
        self.width() <= 0.0 || self.height() <= 0.0
    }
    pub fn normalized(&self) -> bool {
        self.xmin >= -1.0 && self.xmax < 2.0 && self.ymin >= -1.0
    }
}

/// types.rs:176-187
#[derive(Debug, Clone, Copy)]
pub struct Landmark {
    pub x: f64,
    pub y: f64,
    pub z: f64,
}

impl Landmark {
    pub fn new(x: f64, y: f64, z: f64) -> Self {
        Self { x, y, z }
    }
    pub(crate) fn to_mi(self) -> ffi::mi_landmark {
        ffi::mi_landmark { x: self.x, y: self.y, z: self.z }
    }
}

/// types.rs:189-246
#[derive(Debug, Clone)]
pub struct Detection {
    pub data: [[f32; 2]; 8],
    pub score: f32,
}

impl Detection {
    pub(crate) fn from_mi(d: &ffi::mi_detection) -> Detection {
        let mut data = [[0f32; 2]; 8];
        for (k, row) in data.iter_mut().enumerate() {
            row[0] = d.data[2 * k];
            row[1] = d.data[2 * k + 1];
        }
        Detection { data, score: d.score }
    }

    pub(crate) fn to_mi(&self) -> ffi::mi_detection {
        let mut data = [0f32; 16];
        for (k, row) in self.data.iter().enumerate() {
            data[2 * k] = row[0];
            data[2 * k + 1] = row[1];
        }
        ffi::mi_detection { data, score: self.score }
    }

    pub fn keypoint_count(&self) -> usize {
        self.data.len() - 2
    }

    pub fn keypoint(&self, key: usize) -> (f32, f32) {
        (self.data[key + 2][0], self.data[key + 2][1])
    }

    pub fn bbox(&self) -> BBox {
        BBox::new(self.data[0][0] as f64, self.data[0][1] as f64, self.data[1][0] as f64, self.data[1][1] as f64)
    }

    /// types.rs:237-245 (f32 multiplication, as in the reference)
    pub fn scaled_by_image_size(&self, image_size: (i32, i32)) -> Detection {
        let (w, h) = (image_size.0 as f32, image_size.1 as f32);
        let mut data = self.data;
        for row in data.iter_mut() {
            row[0] *= w;
            row[1] *= h;
        }
        Detection { data, score: self.score }
    }
}

/// iris_landmark.rs:115-129 — contour: 71 eye-region points, iris: 5 keypoints
pub struct IrisResults {
    pub(crate) contour: Vec<Landmark>,
    pub(crate) iris: Vec<Landmark>,
}

impl IrisResults {
    pub fn new(contour: Vec<Landmark>, iris: Vec<Landmark>) -> Self {
        Self { contour, iris }
    }
    /// the first MAX_EYE_LANDMARK = 15 contour points (iris_landmark.rs:125-128)
    pub fn eyeball_contour(&self) -> Vec<Landmark> {
        self.contour[0..15].to_vec()
    }
    pub fn contour(&self) -> &[Landmark] {
        &self.contour
    }
    pub fn iris(&self) -> &[Landmark] {
        &self.iris
    }
}

/// Borrowed 8UC3 **RGB** picture — the role `&opencv::core::Mat` plays in the reference (utils.rs:8-21).  `stride` = bytes
/// per row (>= 3 * width); a strided view such as a `Mat` ROI is fine.  The fields are private: the only way to make one is
/// `Image::new` (or a `TryFrom` below), which checks the buffer against the geometry, so the C library never reads past
/// `data` on behalf of safe code.
#[derive(Debug, Clone, Copy)]
pub struct Image<'a> {
    pub(crate) data: &'a [u8],
    pub(crate) width: i32,
    pub(crate) height: i32,
    pub(crate) stride: i32,
}

impl<'a> Image<'a> {
    pub fn new(data: &'a [u8], width: i32, height: i32, stride: i32) -> Result<Image<'a>, anyhow::Error> {
        if width <= 0 || height <= 0 || (stride as i64) < 3 * width as i64 {
            return Err(anyhow::Error::msg("bad image geometry"));
        }
        let need = (height as usize - 1)
            .checked_mul(stride as usize)
            .and_then(|v| v.checked_add(3 * width as usize))
            .ok_or_else(|| anyhow::Error::msg("image geometry overflows"))?;
        if data.len() < need {
            return Err(anyhow::Error::msg("image buffer shorter than its geometry"));
        }
        Ok(Image { data, width, height, stride })
    }
    pub fn data(&self) -> &'a [u8] {
        self.data
    }
    pub fn width(&self) -> i32 {
        self.width
    }
    pub fn height(&self) -> i32 {
        self.height
    }
    pub fn stride(&self) -> i32 {
        self.stride
    }
}

/// `infer(&image, ..)` with an `Image` view: the generic `infer<I: TryInto<Image>>` takes it by reference like a `&Mat`.
impl<'a, 'b> From<&'b Image<'a>> for Image<'a> {
    fn from(v: &'b Image<'a>) -> Image<'a> {
        *v
    }
}

impl<'a> TryFrom<&'a crate::utils::RgbImage> for Image<'a> {
    type Error = anyhow::Error;
    /// the picture `convert_image_to_mat` returns, passed to `infer` by reference as the reference passes its `&Mat`
    fn try_from(m: &'a crate::utils::RgbImage) -> Result<Image<'a>, anyhow::Error> {
        Image::new(m.data(), m.width(), m.height(), 3 * m.width())
    }
}

#[cfg(feature = "opencv")]
impl<'a> TryFrom<&'a opencv::core::Mat> for Image<'a> {
    type Error = anyhow::Error;
    /// `&Mat` as produced by `convert_image_to_mat` (8UC3, RGB order)
    fn try_from(m: &'a opencv::core::Mat) -> Result<Image<'a>, anyhow::Error> {
        use opencv::core::{MatTraitConst, MatTraitConstManual, CV_8UC3};
        if m.typ() != CV_8UC3 {
            return Err(anyhow::Error::msg("expected an 8UC3 Mat"));
        }
        let stride = m.step1(0)? as i32; // elements of depth CV_8U per row = bytes per row
        let bytes = m.data_bytes()?;
        Image::new(bytes, m.cols(), m.rows(), stride)
    }
}
