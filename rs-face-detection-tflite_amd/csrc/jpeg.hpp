// jpeg.hpp — `convert_image_to_mat` (/root/reference/src/face_detection_lite/utils.rs:8-21: cv::imdecode(IMREAD_COLOR) +
// cvtColor(BGR2RGB)) for baseline / extended-sequential and progressive Huffman JPEGs: the serial part (marker parsing, entropy decoding)
// on the host, the sample arithmetic (dequantise + ISLOW IDCT, fancy chroma upsampling, YCbCr -> RGB: libjpeg-turbo's
// algorithms, which is what OpenCV's imdecode runs) on the GPU, leaving RGB u8 in HBM for image_to_tensor / the pipeline.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace mi {

struct JpegComponent {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int bw = 0, bh = 0;   // blocks per row / column (whole MCUs)
    int dw = 0, dh = 0;   // real downsampled size
    size_t coef_off = 0;  // first coefficient of this component in JpegFrame::coef
};

// The coefficient buffer of a frame: the decoder's own vector, or memory the caller provides (`provide`, asked once per picture for `count`
// int16 values: the streamed entries decode straight into pinned host memory, so that the copy to the device is asynchronous).
struct JpegCoefBuf {
    std::vector<int16_t> own;
    int16_t* ext = nullptr;
    size_t n = 0;
    int16_t* (*provide)(size_t count, void* ctx) = nullptr;
    void* ctx = nullptr;
    void assign(size_t count, int16_t v) {
        n = count;
        if (provide) {
            ext = provide(count, ctx);
            for (size_t i = 0; i < count; i++) ext[i] = v;
        } else {
            ext = nullptr;
            own.assign(count, v);
        }
    }
    int16_t* data() { return ext ? ext : own.data(); }
    const int16_t* data() const { return ext ? ext : own.data(); }
    size_t size() const { return n; }
};

struct JpegFrame {             // host-side result of the entropy decoder
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1;
    bool progressive = false;  // SOF2: the coefficients are the sum of several scans (T.81 Annex G)
    JpegComponent comp[3];
    uint16_t qt[4][64] = {};   // natural order
    JpegCoefBuf coef;          // per component [bh][bw][64], natural order, quantised
};

// Parses the headers only. Throws std::runtime_error("unsupported ...") for streams outside the subset.
void jpeg_parse_size(const uint8_t* data, size_t n, int* width, int* height);
// Headers + Huffman decoding: the single interleaved scan of a sequential frame, or every scan of a progressive one.
void jpeg_entropy_decode(const uint8_t* data, size_t n, JpegFrame* out);

// Device side. d_coef: the frame's coefficients (int16), d_qt: uint16 [4][64], d_planes: scratch of jpeg_plane_bytes(f),
// d_rgb: [height][width][3].
size_t jpeg_plane_bytes(const JpegFrame& f);
int launch_jpeg_idct(const JpegFrame& f, const int16_t* d_coef, const uint16_t* d_qt, uint8_t* d_planes, hipStream_t s);
int launch_jpeg_color(const JpegFrame& f, const uint8_t* d_planes, uint8_t* d_rgb, hipStream_t s);

}  // namespace mi
