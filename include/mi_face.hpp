// mi_face.hpp — C++17 host-side mirror of the reference crate's public API over the C ABI (mi_face.h).
//
// The reference is compiled (Rust) code; no Rust toolchain exists in this image, so the host side above the C ABI is
// written in C++ with the reference's names, argument meaning and error behaviour:
//   FaceDetection::{new, infer}   /root/reference/src/face_detection_lite/face_detection.rs:153,205
//   FaceLandmark::{new, infer}    face_landmark.rs:208,232        face_detection_to_roi  face_landmark.rs:180
//   IrisLandmark::{new, infer}    iris_landmark.rs:142,158        iris_roi_from_face_landmarks iris_landmark.rs:268
//   Detection / Rect / Landmark / BBox / IrisResults             types.rs:24-246, iris_landmark.rs:115-129
// `anyhow::Error` (and the reference's panics) become `mi_face::Error`; `Option<T>` becomes `std::optional<T>`;
// `&Mat` (8UC3 RGB, utils.rs:8-21) becomes the non-owning `Image` view.  Header-only; link with -lmiface.
#pragma once

#include <array>
#include <algorithm>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mi_face.h"

namespace mi_face {

class Error : public std::runtime_error {
   public:
    Error(int code, const std::string& msg) : std::runtime_error(msg), code_(code) {}
    int code() const { return code_; }

   private:
    int code_;
};

namespace detail {
inline void check(int rc) {
    if (rc != MI_OK) throw Error(rc, mi_last_error());
}
}  // namespace detail

// types.rs:24-36
struct Rect {
    double x_center = 0.5, y_center = 0.5, width = 1.0, height = 1.0, rotation = 0.0;
    bool normalized = true;
    mi_rect c() const { return mi_rect{x_center, y_center, width, height, rotation, normalized ? 1 : 0}; }
    static Rect from(const mi_rect& r) { return Rect{r.x_center, r.y_center, r.width, r.height, r.rotation, r.normalized != 0}; }
};

// types.rs:99-174 (the parts callers use)
struct BBox {
    double xmin, ymin, xmax, ymax;
    double width() const { return xmax - xmin; }
    double height() const { return ymax - ymin; }
};

// types.rs:176-187
struct Landmark {
    double x, y, z;
};

// types.rs:189-246: data is [8][2] = (xmin,ymin), (xmax,ymax), 6 keypoints
struct Detection {
    std::array<float, 16> data{};
    float score = 0.f;
    std::size_t keypoint_count() const { return 6; }
    std::pair<float, float> keypoint(std::size_t k) const { return {data[2 * (k + 2)], data[2 * (k + 2) + 1]}; }
    BBox bbox() const { return BBox{data[0], data[1], data[2], data[3]}; }
};

// iris_landmark.rs:115-129
struct IrisResults {
    std::vector<Landmark> contour;  // 71
    std::vector<Landmark> iris;     // 5
    std::vector<Landmark> eyeball_contour() const { return std::vector<Landmark>(contour.begin(), contour.begin() + 15); }
};

// face_detection.rs:117-123
enum class FaceDetectionModel { FrontCamera = 0, BackCamera = 1, Short = 2, Full = 3, FullSparse = 4 };

// Stand-in for `&opencv::core::Mat` holding 8UC3 RGB pixels (utils.rs:8-21); does not own the pixels.
struct Image {
    const std::uint8_t* rgb;
    int width, height;
    int stride;  // bytes per row
};

class FaceDetection {
   public:
    // FaceDetection::new(model_type, model_path): model_path is a DIRECTORY, default "./models" (face_detection.rs:157-161)
    explicit FaceDetection(FaceDetectionModel model_type, std::optional<std::string> model_path = std::nullopt, int device = 0) {
        detail::check(mi_fd_create(static_cast<int>(model_type), model_path ? model_path->c_str() : nullptr, device, &h_));
    }
    ~FaceDetection() { mi_fd_free(h_); }
    FaceDetection(const FaceDetection&) = delete;
    FaceDetection& operator=(const FaceDetection&) = delete;

    // FaceDetection::infer(&Mat, Option<Rect>) -> Vec<Detection> (face_detection.rs:205-267)
    std::vector<Detection> infer(const Image& image, std::optional<Rect> roi = std::nullopt) const {
        std::vector<mi_detection> out(256);
        int n = 0;
        mi_rect r{};
        if (roi) r = roi->c();
        for (;;) {  // the reference returns every detection: when more were found than fit, ask again with that capacity
            detail::check(mi_fd_infer_image(h_, image.rgb, image.width, image.height, image.stride, roi ? &r : nullptr, out.data(),
                                            static_cast<int>(out.size()), &n));
            if (n <= static_cast<int>(out.size())) break;
            out.resize(static_cast<std::size_t>(n));
        }
        std::vector<Detection> dets(static_cast<std::size_t>(n));
        for (std::size_t i = 0; i < dets.size(); i++) {
            for (int k = 0; k < 16; k++) dets[i].data[k] = out[i].data[k];
            dets[i].score = out[i].score;
        }
        return dets;
    }
    // infer() over a batch of equally sized frames in one call (mi_fd_infer_images): `frames` = batch x height rows of `stride`
    // bytes, host memory; rois empty (whole frames) or one per frame.  Up to cap_per_frame detections per frame are returned.
    std::vector<std::vector<Detection>> infer_batch(const std::uint8_t* frames, int batch, int width, int height, int stride,
                                                    const std::vector<Rect>& rois = {}, int cap_per_frame = 64) const {
        if (batch < 1 || cap_per_frame < 1) throw std::invalid_argument("infer_batch: batch and cap_per_frame must be positive");
        if (!rois.empty() && rois.size() != static_cast<std::size_t>(batch))   // (the C side reads `batch` entries)
            throw std::invalid_argument("infer_batch: rois must be empty or hold one entry per frame");
        std::vector<mi_detection> out(static_cast<std::size_t>(batch) * cap_per_frame);
        std::vector<int> counts(static_cast<std::size_t>(batch));
        std::vector<mi_rect> r;
        for (const Rect& x : rois) r.push_back(x.c());
        detail::check(mi_fd_infer_images(h_, frames, batch, width, height, stride, r.empty() ? nullptr : r.data(), out.data(), cap_per_frame,
                                         counts.data(), MI_MEM_HOST, nullptr));
        std::vector<std::vector<Detection>> res(static_cast<std::size_t>(batch));
        for (int b = 0; b < batch; b++)
            for (int i = 0; i < std::min(counts[static_cast<std::size_t>(b)], cap_per_frame); i++) {
                Detection d;
                const mi_detection& m = out[static_cast<std::size_t>(b) * cap_per_frame + i];
                for (int k = 0; k < 16; k++) d.data[k] = m.data[k];
                d.score = m.score;
                res[static_cast<std::size_t>(b)].push_back(d);
            }
        return res;
    }
    // convert_image_to_mat + infer for a stream of encoded pictures (utils.rs:8-21, face_detection.rs:205): submit_jpeg decodes the entropy-coded
    // data on this thread while the device still runs the other slot's picture and queues the rest; collect_jpeg returns that slot's detections
    void submit_jpeg(int slot, const std::uint8_t* bytes, std::size_t nbytes, int cap = 64) const { detail::check(mi_fd_submit_jpeg(h_, slot, bytes, nbytes, cap)); }
    std::vector<Detection> collect_jpeg(int slot, int cap = 64, int* width = nullptr, int* height = nullptr) const {
        std::vector<mi_detection> out(static_cast<std::size_t>(cap));
        int n = 0;
        detail::check(mi_fd_collect_jpeg(h_, slot, out.data(), cap, &n, width, height));
        std::vector<Detection> dets(static_cast<std::size_t>(std::min(n, cap)));
        for (std::size_t i = 0; i < dets.size(); i++) {
            for (int k = 0; k < 16; k++) dets[i].data[k] = out[i].data[k];
            dets[i].score = out[i].score;
        }
        return dets;
    }
    mi_fd* handle() const { return h_; }

   private:
    mi_fd* h_ = nullptr;
};

class FaceLandmark {
   public:
    // FaceLandmark::new(model_path): FILE path, default "./models/face_landmark.tflite" (face_landmark.rs:211-215)
    explicit FaceLandmark(std::optional<std::string> model_path = std::nullopt, int device = 0) {
        detail::check(mi_fl_create(model_path ? model_path->c_str() : nullptr, device, &h_));
    }
    ~FaceLandmark() { mi_fl_free(h_); }
    FaceLandmark(const FaceLandmark&) = delete;
    FaceLandmark& operator=(const FaceLandmark&) = delete;

    // FaceLandmark::infer(&Mat, Option<Rect>) -> Vec<Landmark>; empty when the face flag fails (face_landmark.rs:292-296)
    std::vector<Landmark> infer(const Image& image, std::optional<Rect> roi = std::nullopt) const {
        std::vector<mi_landmark> out(MI_NUM_FACE_LANDMARKS);
        int n = 0;
        mi_rect r{};
        if (roi) r = roi->c();
        detail::check(mi_fl_infer_image(h_, image.rgb, image.width, image.height, image.stride, roi ? &r : nullptr, out.data(),
                                        MI_NUM_FACE_LANDMARKS, &n));
        std::vector<Landmark> lm(static_cast<std::size_t>(n));
        for (std::size_t i = 0; i < lm.size(); i++) lm[i] = Landmark{out[i].x, out[i].y, out[i].z};
        return lm;
    }

    // infer() over a batch (mi_fl_infer_images): `frames` = batch x height rows of `stride` bytes, host memory; item i reads frame
    // i / items_per_frame with ROI rois[i] (rois empty: whole frames, items_per_frame 1).  One entry per item; empty where the
    // face flag fails, like infer().
    std::vector<std::vector<Landmark>> infer_batch(const std::uint8_t* frames, int batch, int width, int height, int stride,
                                                   const std::vector<Rect>& rois = {}, int items_per_frame = 1) const {
        if (batch < 1 || items_per_frame < 1) throw std::invalid_argument("infer_batch: batch and items_per_frame must be positive");
        const std::size_t n = static_cast<std::size_t>(batch) * items_per_frame;
        if (rois.empty() ? items_per_frame != 1 : rois.size() != n) throw std::invalid_argument("infer_batch: rois must hold one entry per item");
        std::vector<mi_rect> r;
        for (const Rect& x : rois) r.push_back(x.c());
        std::vector<float> lm(n * 3 * MI_NUM_FACE_LANDMARKS);
        std::vector<int> present(n);
        detail::check(mi_fl_infer_images(h_, frames, batch, width, height, stride, r.empty() ? nullptr : r.data(), items_per_frame, lm.data(),
                                         present.data(), nullptr, MI_MEM_HOST, nullptr));
        std::vector<std::vector<Landmark>> res(n);
        for (std::size_t i = 0; i < n; i++)
            if (present[i])
                for (int k = 0; k < MI_NUM_FACE_LANDMARKS; k++) {
                    const float* v = &lm[(i * MI_NUM_FACE_LANDMARKS + k) * 3];
                    res[i].push_back(Landmark{v[0], v[1], v[2]});
                }
        return res;
    }

   private:
    mi_fl* h_ = nullptr;
};

class IrisLandmark {
   public:
    explicit IrisLandmark(std::optional<std::string> model_path = std::nullopt, int device = 0) {
        detail::check(mi_iris_create(model_path ? model_path->c_str() : nullptr, device, &h_));
    }
    ~IrisLandmark() { mi_iris_free(h_); }
    IrisLandmark(const IrisLandmark&) = delete;
    IrisLandmark& operator=(const IrisLandmark&) = delete;

    // IrisLandmark::infer(&Mat, Option<Rect>, Option<bool>) -> IrisResults (iris_landmark.rs:158-248)
    IrisResults infer(const Image& image, std::optional<Rect> roi = std::nullopt, std::optional<bool> is_right_eye = std::nullopt) const {
        mi_landmark c[MI_NUM_EYE_LANDMARKS], i5[MI_NUM_IRIS_LANDMARKS];
        mi_rect r{};
        if (roi) r = roi->c();
        detail::check(mi_iris_infer_image(h_, image.rgb, image.width, image.height, image.stride, roi ? &r : nullptr,
                                          is_right_eye.value_or(false) ? 1 : 0, c, i5));
        IrisResults res;
        for (const auto& v : c) res.contour.push_back(Landmark{v.x, v.y, v.z});
        for (const auto& v : i5) res.iris.push_back(Landmark{v.x, v.y, v.z});
        return res;
    }

    // infer() over a batch (mi_iris_infer_images): item i reads frame i / items_per_frame with ROI rois[i] and flip is_right_eye[i]
    // (empty: no flips); rois empty: whole frames, items_per_frame 1.
    std::vector<IrisResults> infer_batch(const std::uint8_t* frames, int batch, int width, int height, int stride, const std::vector<Rect>& rois = {},
                                         const std::vector<bool>& is_right_eye = {}, int items_per_frame = 1) const {
        if (batch < 1 || items_per_frame < 1) throw std::invalid_argument("infer_batch: batch and items_per_frame must be positive");
        const std::size_t n = static_cast<std::size_t>(batch) * items_per_frame;
        if (rois.empty() ? items_per_frame != 1 : rois.size() != n) throw std::invalid_argument("infer_batch: rois must hold one entry per item");
        if (!is_right_eye.empty() && is_right_eye.size() != n) throw std::invalid_argument("infer_batch: is_right_eye must hold one entry per item");
        std::vector<mi_rect> r;
        for (const Rect& x : rois) r.push_back(x.c());
        std::vector<int> flip(is_right_eye.begin(), is_right_eye.end());
        std::vector<float> c(n * 3 * MI_NUM_EYE_LANDMARKS), i5(n * 3 * MI_NUM_IRIS_LANDMARKS);
        detail::check(mi_iris_infer_images(h_, frames, batch, width, height, stride, r.empty() ? nullptr : r.data(), flip.empty() ? nullptr : flip.data(),
                                           items_per_frame, c.data(), i5.data(), MI_MEM_HOST, nullptr));
        std::vector<IrisResults> res(n);
        for (std::size_t i = 0; i < n; i++) {
            for (int k = 0; k < MI_NUM_EYE_LANDMARKS; k++) res[i].contour.push_back(Landmark{c[(i * MI_NUM_EYE_LANDMARKS + k) * 3], c[(i * MI_NUM_EYE_LANDMARKS + k) * 3 + 1], c[(i * MI_NUM_EYE_LANDMARKS + k) * 3 + 2]});
            for (int k = 0; k < MI_NUM_IRIS_LANDMARKS; k++) res[i].iris.push_back(Landmark{i5[(i * MI_NUM_IRIS_LANDMARKS + k) * 3], i5[(i * MI_NUM_IRIS_LANDMARKS + k) * 3 + 1], i5[(i * MI_NUM_IRIS_LANDMARKS + k) * 3 + 2]});
        }
        return res;
    }

   private:
    mi_iris* h_ = nullptr;
};

// face_detection_to_roi(face_detection, image_size, None) (face_landmark.rs:180-198); image_size = (width, height)
inline Rect face_detection_to_roi(const Detection& d, std::pair<int, int> image_size) {
    mi_detection c{};
    for (int k = 0; k < 16; k++) c.data[k] = d.data[k];
    c.score = d.score;
    mi_rect r{};
    detail::check(mi_face_detection_to_roi(&c, image_size.first, image_size.second, &r));
    return Rect::from(r);
}

// iris_roi_from_face_landmarks(face_landmarks, image_size) -> (left_eye_roi, right_eye_roi) (iris_landmark.rs:268-292)
inline std::pair<Rect, Rect> iris_roi_from_face_landmarks(const std::vector<Landmark>& lm, std::pair<int, int> image_size) {
    if (lm.size() < MI_NUM_FACE_LANDMARKS) throw Error(MI_EINVAL, "expected 468 face landmarks");  // the reference would panic on lm[362]
    std::vector<mi_landmark> c(MI_NUM_FACE_LANDMARKS);
    for (int i = 0; i < MI_NUM_FACE_LANDMARKS; i++) c[i] = mi_landmark{lm[i].x, lm[i].y, lm[i].z};
    mi_rect l{}, r{};
    detail::check(mi_iris_roi_from_face_landmarks(c.data(), image_size.first, image_size.second, &l, &r));
    return {Rect::from(l), Rect::from(r)};
}

// bbox_to_roi(bbox, image_size, rotation_keypoints, scale, mode) (transform.rs:44-109); SizeMode as in transform.rs:24-34
enum class SizeMode { Default = 0, SquareLong = 1, SquareShort = 2 };
inline Rect bbox_to_roi(const std::array<double, 4>& bbox, std::pair<int, int> image_size,
                        std::optional<std::array<double, 4>> rotation_keypoints = std::nullopt, std::pair<double, double> scale = {1.0, 1.0},
                        SizeMode mode = SizeMode::Default) {
    mi_rect r{};
    detail::check(mi_bbox_to_roi(bbox.data(), image_size.first, image_size.second, rotation_keypoints ? rotation_keypoints->data() : nullptr,
                                 scale.first, scale.second, static_cast<int>(mode), &r));
    return Rect::from(r);
}
// bbox_from_landmarks(landmarks) (transform.rs:146-165) -> {xmin, ymin, xmax, ymax}
inline std::array<double, 4> bbox_from_landmarks(const std::vector<Landmark>& lm) {
    std::vector<mi_landmark> c(lm.size());
    for (size_t i = 0; i < lm.size(); i++) c[i] = mi_landmark{lm[i].x, lm[i].y, lm[i].z};
    std::array<double, 4> out{};
    detail::check(mi_bbox_from_landmarks(c.data(), static_cast<int>(c.size()), out.data()));
    return out;
}

// convert_image_to_mat(im_bytes) (utils.rs:8-21): an owning RGB picture; `.image()` is the borrowed view the infer() calls take
struct OwnedImage {
    std::vector<std::uint8_t> rgb;
    int width = 0, height = 0;
    Image image() const { return Image{rgb.data(), width, height, 3 * width}; }
};
inline OwnedImage convert_image_to_mat(const std::uint8_t* im_bytes, std::size_t n, int device = 0) {
    OwnedImage o;
    detail::check(mi_jpeg_info(im_bytes, n, &o.width, &o.height));
    o.rgb.resize(static_cast<std::size_t>(3) * o.width * o.height);
    detail::check(mi_jpeg_decode_rgb(device, im_bytes, n, o.rgb.data(), o.rgb.size(), &o.width, &o.height, MI_MEM_HOST, nullptr));
    return o;
}

// update_face_landmarks_with_iris_results(face_landmarks, iris_data_left, iris_data_right) (iris_landmark.rs:380-398)
inline std::vector<Landmark> update_face_landmarks_with_iris_results(const std::vector<Landmark>& face_landmarks, const IrisResults& left,
                                                                     const IrisResults& right) {
    if (face_landmarks.size() != MI_NUM_FACE_LANDMARKS) throw Error(MI_EINVAL, "unexpected number of items in face_landmarks");  // the reference's Err
    if (left.contour.size() != MI_NUM_EYE_LANDMARKS || right.contour.size() != MI_NUM_EYE_LANDMARKS) throw Error(MI_EINVAL, "expected 71 contour landmarks per eye");
    auto pack = [](const std::vector<Landmark>& v) {
        std::vector<mi_landmark> c(v.size());
        for (size_t i = 0; i < v.size(); i++) c[i] = mi_landmark{v[i].x, v[i].y, v[i].z};
        return c;
    };
    std::vector<mi_landmark> f = pack(face_landmarks), l = pack(left.contour), r = pack(right.contour);
    detail::check(mi_update_face_landmarks_with_iris_results(f.data(), l.data(), r.data(), f.data()));
    std::vector<Landmark> out(f.size());
    for (size_t i = 0; i < f.size(); i++) out[i] = Landmark{f[i].x, f[i].y, f[i].z};
    return out;
}

}  // namespace mi_face
