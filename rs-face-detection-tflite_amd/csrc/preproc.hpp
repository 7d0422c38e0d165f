// preproc.hpp — device-side transform::image_to_tensor (/root/reference/src/face_detection_lite/transform.rs:188-309),
// for one image or for a batch of (frame, ROI) items whose ROIs already live in device memory.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/mi_face.h"
#include "kernels.hpp"

namespace mi {

// Everything image_to_tensor derives from (image size, ROI, output size, keep_aspect): computed once per item, on the
// host for the single-image entry points and by pre_geom_kernel on the device for batches (same code).
struct PreGeom {
    double Minv[9];             // dst -> src homography of the ROI warp (cv::getPerspectiveTransform + invert)
    double pad_x, pad_y;        // ImageTensor.padding = (pad_x, pad_y, pad_x, pad_y)
    int warp_w, warp_h, bw0;    // warpPerspective target size, OpenCV tile width
    int stage1;                 // 0: none, 1: border only (same-size resize = copy), 2: border + resize to (new_w, new_h)
    int pad_h, pad_v, new_w, new_h;
    int stage2;                 // 1: resize to the output size, 0: already there
    int valid;                  // 0: item skipped (no face / degenerate ROI): output tensor is zero-filled
};

struct PreItems {               // batch description (device pointers unless stated)
    const uint8_t* frames;      // [n_frames][height][stride] RGB u8
    long frame_bytes;           // bytes between frames
    int width, height, stride;
    const RectD* rois;          // [N] or null (whole image)
    const int* roi_valid;       // [N] or null
    const int* flip;            // [N] or null
    int items_per_frame;        // item i reads frame i / items_per_frame
    int N, out_w, out_h, keep_aspect;
    double range_min, range_max;
};

// geometry for N items -> d_geom[N], d_padding[N][4] (may be null)
void launch_pre_geom(const PreItems& it, PreGeom* d_geom, double* d_padding, hipStream_t s);
// ... of N whole pictures of one size (no ROI), computed on the host and uploaded (synchronous: the caller caches it per (width, height, N))
void upload_whole_image_geom(int width, int height, int out_w, int out_h, bool keep_aspect, int N, PreGeom* d_geom, double* d_padding, hipStream_t s);
// warp -> (border, resize) -> resize -> flip -> normalise, fused per output pixel; out f32 [N][out_h][out_w][3]
void launch_pre_tensor(const PreItems& it, const PreGeom* d_geom, float* d_out, hipStream_t s);

// Single-image convenience used by the mi_*_infer_image entry points: uploads the host image into d_scratch
// (>= stride*height + 256 bytes + sizeof(PreGeom)), runs the two kernels with N = 1, returns the padding on the host.
void image_to_tensor_device(const uint8_t* rgb_host, int width, int height, int stride, const mi_rect* roi, int out_w, int out_h,
                            bool keep_aspect_ratio, double range_min, double range_max, bool flip_horizontal, float* d_out,
                            double padding[4], void* d_scratch, hipStream_t stream);
// The same without a host synchronisation and without small uploads: geometry and flip flag travel as kernel arguments, the picture
// is copied into d_img (>= stride * height bytes) on `stream`.  `rgb_host` must stay valid until the caller has synchronised `stream`.
void image_to_tensor_enqueue(const uint8_t* rgb_host, int width, int height, int stride, const mi_rect* roi, int out_w, int out_h,
                             bool keep_aspect_ratio, double range_min, double range_max, bool flip_horizontal, float* d_out,
                             double padding[4], uint8_t* d_img, hipStream_t stream);
// ... for a picture that is in device memory already (the streamed JPEG entries: the decoder's RGB output never visits the host)
void image_to_tensor_enqueue_device(const uint8_t* d_img, int width, int height, int stride, const mi_rect* roi, int out_w, int out_h,
                                    bool keep_aspect_ratio, double range_min, double range_max, bool flip_horizontal, float* d_out,
                                    double padding[4], hipStream_t stream);
size_t image_to_tensor_scratch_bytes(int width, int height, int stride, const mi_rect* roi, int out_w, int out_h, bool keep_aspect_ratio);

// Device-side ROI maths between pipeline stages (face_landmark.rs:180-198, iris_landmark.rs:268-292).
void launch_face_rois(const float* d_dets /*[B][cap][17]*/, const int* d_counts, int B, int cap, int image_w, int image_h,
                      RectD* d_rois, int* d_valid, hipStream_t s);
void launch_iris_rois(const float* d_landmarks /*[B][468][3]*/, const int* d_present, int B, int image_w, int image_h,
                      RectD* d_rois /*[B][2] left,right*/, int* d_valid /*[B][2]*/, int* d_flip /*[B][2]*/, hipStream_t s);

}  // namespace mi
