// roi_dev.hpp — ROI maths on the device, shared by preproc.hip (face_roi_kernel / iris_roi_kernel) and kernels.hip (the post-processing kernel
// of the batched pipeline computes faces[0]'s ROI itself).
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace mi {

// bbox_to_roi + select_roi_size(SquareLong) (transform.rs:44-109), f64 like the reference
__device__ inline bool bbox_to_roi_dev(const double bbox[4], int image_w, int image_h, const double kp[4], double scale, RectD* out) {
    const double xmin = bbox[0], ymin = bbox[1], xmax = bbox[2], ymax = bbox[3];
    if (!(xmin >= -1.0 && xmax < 2.0 && ymin >= -1.0)) return false;  // BBox::normalized (types.rs:133-135)
    const double iw = image_w, ih = image_h;
    const double aw = xmax * iw - xmin * iw, ah = ymax * ih - ymin * ih;
    const double side = fmax(aw, ah);
    const double width = side / iw * scale, height = side / ih * scale;
    const double pi = 3.14159265358979323846;
    const double angle = -atan2(kp[1] - kp[3], kp[2] - kp[0]);
    out->x_center = xmin + (xmax - xmin) / 2.0;
    out->y_center = ymin + (ymax - ymin) / 2.0;
    out->width = width;
    out->height = height;
    out->rotation = angle - 2.0 * pi * floor((angle + pi) / (2.0 * pi));
    out->normalized = 1;
    out->pad_ = 0;
    return true;
}

// face_detection_to_roi of faces[0] (lib.rs:29, face_landmark.rs:30,189): d = the frame's first detection (17 floats), count = its detections
__device__ inline int face_roi_dev(const float* d, int count, int image_w, int image_h, RectD* out) {
    RectD r = {0.5, 0.5, 1.0, 1.0, 0.0, 1, 0};
    int ok = 0;
    if (count > 0) {  // Detection::scaled_by_image_size multiplies in f32 (types.rs:237-245)
        const float w = (float)image_w, h = (float)image_h;
        const double kp[4] = {(double)__fmul_rn(d[4], w), (double)__fmul_rn(d[5], h), (double)__fmul_rn(d[6], w), (double)__fmul_rn(d[7], h)};
        const double bbox[4] = {(double)d[0], (double)d[1], (double)d[2], (double)d[3]};
        ok = bbox_to_roi_dev(bbox, image_w, image_h, kp, 1.5, &r) ? 1 : 0;  // ROI_SCALE, SquareLong
    }
    *out = r;
    return ok;
}

}  // namespace mi
