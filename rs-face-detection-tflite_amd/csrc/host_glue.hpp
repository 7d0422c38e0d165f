// host_glue.hpp — the small host-side pieces of the reference crate that stay on the CPU: SSD anchor generation
// (runs once per handle) and the ROI maths between pipeline stages (a few flops per face).
#pragma once

#include <vector>

#include "../../include/mi_face.h"

namespace mi {

struct SsdOptions {  // SSDOptions, face_detection.rs:28-86
    int num_layers, input_h, input_w;
    float anchor_offset_x, anchor_offset_y;
    int strides[4];
    float interpolated_scale_aspect_ratio;
};

// SSDOptions::new_front/new_back/new_short/new_full selected as FaceDetection::new does (face_detection.rs:163-185).
bool ssd_options_for(int kind, SsdOptions* out);
const char* model_file_for(int kind);  // MODEL_NAME_* (face_detection.rs:125-129)
// ssd_generate_anchors (face_detection.rs:366-413): [n][2] (x_center, y_center), f32.
std::vector<float> ssd_generate_anchors(const SsdOptions& o);

// bbox_to_roi + select_roi_size (transform.rs:44-109). size_mode: 0 Default, 1 SquareLong, 2 SquareShort.
bool bbox_to_roi(const double bbox[4], int image_w, int image_h, const double* rotation_keypoints /*x0,y0,x1,y1 or null*/,
                 double scale_x, double scale_y, int size_mode, mi_rect* out);

}  // namespace mi
