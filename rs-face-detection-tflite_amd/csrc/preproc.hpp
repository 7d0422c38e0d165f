// preproc.hpp — device-side transform::image_to_tensor (/root/reference/src/face_detection_lite/transform.rs:188-309).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/mi_face.h"

namespace mi {

// Runs ROI warp -> (letterbox pad + resize) -> resize -> flip -> normalise on `stream`.
//   rgb_host   8UC3 RGB rows of `stride` bytes in HOST memory (uploaded here)
//   d_out      DEVICE f32 [out_h][out_w][3]
//   padding    host out: (left, top, right, bottom) = ImageTensor.padding (transform.rs:306)
//   d_scratch  DEVICE scratch of at least image_to_tensor_scratch_bytes(...) bytes
void image_to_tensor_device(const uint8_t* rgb_host, int width, int height, int stride, const mi_rect* roi, int out_w, int out_h,
                            bool keep_aspect_ratio, double range_min, double range_max, bool flip_horizontal, float* d_out,
                            double padding[4], void* d_scratch, hipStream_t stream);
size_t image_to_tensor_scratch_bytes(int width, int height, int stride, const mi_rect* roi, int out_w, int out_h, bool keep_aspect_ratio);

}  // namespace mi
