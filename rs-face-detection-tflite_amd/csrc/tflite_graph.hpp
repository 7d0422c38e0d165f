// tflite_graph.hpp — dependency-free TFL3 flatbuffer reader -> graph IR (host side, C++17).
//
// Replaces `FlatBufferModel::build_from_file` of the third-party `tflite` crate as used by the reference at
// /root/reference/src/face_detection_lite/face_detection.rs:188, face_landmark.rs:216, iris_landmark.rs:150.
// All constants are widened to f32 at load (the detector graphs store f16 weights behind DEQUANTIZE ops, which
// the reference re-executes on every infer call because it rebuilds its interpreter, face_detection.rs:207-210).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace mi {

enum class BuiltinOp : int {
    Add = 0, Concatenation = 2, Conv2D = 3, DepthwiseConv2D = 4, DepthToSpace = 5, Dequantize = 6, MaxPool2D = 17,
    Relu = 19, Reshape = 22, ResizeBilinear = 23, Pad = 34, Prelu = 54, Densify = 124,
};

enum class Padding : int { Same = 0, Valid = 1 };
enum class FusedAct : int { None = 0, Relu = 1, Relu6 = 3 };

struct TensorInfo {
    std::vector<int> shape;
    int dtype = 0;                 // TFLite TensorType: 0 f32, 1 f16, 2 i32, ...
    bool is_const = false;
    std::vector<float> f32;        // constant payload widened to f32 (dense)
    std::vector<int32_t> i32;      // constant payload for i32 tensors (paddings, sizes)
    std::string name;
    size_t elems() const {
        size_t n = 1;
        for (int d : shape) n *= static_cast<size_t>(d);
        return n;
    }
};

struct OpInfo {
    BuiltinOp op{};
    int raw_code = -1;
    std::vector<int> inputs, outputs;
    Padding padding = Padding::Same;
    int stride_w = 1, stride_h = 1, filter_w = 1, filter_h = 1, depth_multiplier = 1, axis = 0, block_size = 1;
    FusedAct act = FusedAct::None;
    bool align_corners = false, half_pixel_centers = false;
};

struct Graph {
    std::string description;
    std::vector<TensorInfo> tensors;
    std::vector<OpInfo> ops;       // DEQUANTIZE / DENSIFY of constants already folded away
    std::vector<int> inputs, outputs;
};

// Throws std::runtime_error with a readable message on malformed / unsupported input.
Graph parse_tflite(const uint8_t* data, size_t size);

}  // namespace mi
