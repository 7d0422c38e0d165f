// host_glue.cpp — see host_glue.hpp.  Must be built with -ffp-contract=off: the Rust source rounds every
// multiply and add separately.
#include "host_glue.hpp"

#include <algorithm>
#include <cmath>

namespace mi {

bool ssd_options_for(int kind, SsdOptions* out) {
    switch (kind) {
        case MI_FD_FRONT_CAMERA:
        case MI_FD_SHORT:
            *out = SsdOptions{4, 128, 128, 0.5f, 0.5f, {8, 16, 16, 16}, 1.0f};
            return true;
        case MI_FD_BACK_CAMERA:
            *out = SsdOptions{4, 256, 256, 0.5f, 0.5f, {16, 32, 32, 32}, 1.0f};
            return true;
        case MI_FD_FULL:
        case MI_FD_FULL_SPARSE:
            *out = SsdOptions{1, 192, 192, 0.5f, 0.5f, {4, 0, 0, 0}, 0.0f};
            return true;
        default:
            return false;
    }
}

const char* model_file_for(int kind) {
    switch (kind) {
        case MI_FD_FRONT_CAMERA: return "face_detection_front.tflite";
        case MI_FD_BACK_CAMERA: return "face_detection_back.tflite";
        case MI_FD_SHORT: return "face_detection_short_range.tflite";
        case MI_FD_FULL: return "face_detection_full_range.tflite";
        case MI_FD_FULL_SPARSE: return "face_detection_full_range_sparse.tflite";
        default: return nullptr;
    }
}

std::vector<float> ssd_generate_anchors(const SsdOptions& o) {
    std::vector<float> anchors;
    for (int layer = 0; layer < o.num_layers;) {
        // consecutive layers with the same stride share one feature map and stack their anchors per cell
        int next = layer, per_cell = 0;
        for (; next < o.num_layers && o.strides[next] == o.strides[layer]; ++next)
            per_cell += o.interpolated_scale_aspect_ratio == 1.0f ? 2 : 1;
        const int stride = o.strides[layer];
        const int rows = o.input_h / stride, cols = o.input_w / stride;
        for (int y = 0; y < rows; ++y) {
            const float yc = (static_cast<float>(y) + o.anchor_offset_y) / static_cast<float>(rows);
            for (int x = 0; x < cols; ++x) {
                const float xc = (static_cast<float>(x) + o.anchor_offset_x) / static_cast<float>(cols);
                for (int k = 0; k < per_cell; ++k) {
                    anchors.push_back(xc);
                    anchors.push_back(yc);
                }
            }
        }
        layer = next;
    }
    return anchors;
}

bool bbox_to_roi(const double bbox[4], int image_w, int image_h, const double* kp, double scale_x, double scale_y,
                 int size_mode, mi_rect* out) {
    const double xmin = bbox[0], ymin = bbox[1], xmax = bbox[2], ymax = bbox[3];
    if (!(xmin >= -1.0 && xmax < 2.0 && ymin >= -1.0)) return false;  // BBox::normalized (types.rs:133-135)
    const double iw = image_w, ih = image_h;
    // BBox::absolute (normalised boxes are scaled by the image size) then width()/height()
    double width = xmax * iw - xmin * iw, height = ymax * ih - ymin * ih;
    if (size_mode == 1) {
        const double side = std::fmax(width, height);
        width = side / iw;
        height = side / ih;
    } else if (size_mode == 2) {
        const double side = std::fmin(width, height);
        width = side / iw;
        height = side / ih;
    }
    width = width * scale_x;
    height = height * scale_y;
    const double cx = xmin + (xmax - xmin) / 2.0, cy = ymin + (ymax - ymin) / 2.0;
    double rotation = 0.0;
    if (kp) {
        const double pi = 3.14159265358979323846;
        const double angle = -std::atan2(kp[1] - kp[3], kp[2] - kp[0]);
        rotation = angle - 2.0 * pi * std::floor((angle + pi) / (2.0 * pi));
    }
    out->x_center = cx;
    out->y_center = cy;
    out->width = width;
    out->height = height;
    out->rotation = rotation;
    out->normalized = 1;
    return true;
}

}  // namespace mi
