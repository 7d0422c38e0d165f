// preproc.hip — transform::image_to_tensor on the GPU (reference: /root/reference/src/face_detection_lite/
// transform.rs:188-309).  The reference chains OpenCV calls (getPerspectiveTransform, warpPerspective INTER_LINEAR /
// BORDER_CONSTANT 0, copyMakeBorder, resize INTER_LINEAR, flip) and a per-pixel u8 -> f32 loop on the CPU; here the
// same stages run as three small kernels that keep OpenCV's generic 8-bit fixed-point arithmetic
// (1/32-pixel source coordinates + 15-bit weights for the warp, 11-bit coefficient pairs for the resize), so results
// stay within 1 LSB of the u8 intermediate images of a non-IPP OpenCV build.
#include "preproc.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

#include "engine.hpp"

namespace mi {
namespace {

struct Geom {
    double roi_w, roi_h;            // absolute ROI size (untruncated)
    int warp_w, warp_h;             // warpPerspective target size
    double Minv[9];                 // dst -> src homography
    bool two_stage;                 // letterbox: pad + resize to (new_w,new_h)
    int pad_h, pad_v, new_w, new_h;
    double pad_x, pad_y;
};

// 8x8 solve for the 4-point homography (Gauss-Jordan with partial pivoting, f64) and 3x3 inverse.
bool homography(const float src[4][2], const float dst[4][2], double M[9]) {
    double a[8][9];
    std::memset(a, 0, sizeof a);
    for (int i = 0; i < 4; i++) {
        const double X = src[i][0], Y = src[i][1], u = dst[i][0], v = dst[i][1];
        double r0[9] = {X, Y, 1, 0, 0, 0, -X * u, -Y * u, u};
        double r1[9] = {0, 0, 0, X, Y, 1, -X * v, -Y * v, v};
        std::memcpy(a[i], r0, sizeof r0);
        std::memcpy(a[i + 4], r1, sizeof r1);
    }
    for (int c = 0; c < 8; c++) {
        int p = c;
        for (int r = c + 1; r < 8; r++)
            if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
        if (std::fabs(a[p][c]) < 1e-300) return false;
        if (p != c) std::swap_ranges(a[c], a[c] + 9, a[p]);
        for (int r = 0; r < 8; r++) {
            if (r == c) continue;
            const double f = a[r][c] / a[c][c];
            if (f != 0.0)
                for (int k = c; k < 9; k++) a[r][k] -= f * a[c][k];
        }
    }
    for (int i = 0; i < 8; i++) M[i] = a[i][8] / a[i][i];
    M[8] = 1.0;
    return true;
}

bool inverse3(const double m[9], double o[9]) {
    const double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    o[0] = c0 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c1 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c2 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

Geom make_geom(int width, int height, const mi_rect* roi_in, int out_w, int out_h, bool keep_aspect) {
    Geom g{};
    mi_rect roi = roi_in ? *roi_in : mi_rect{0.5, 0.5, 1.0, 1.0, 0.0, 1};  // transform.rs:190-197
    if (roi.normalized) {                                                   // Rect::scaled(size, false), types.rs:62-77
        roi.x_center *= width; roi.y_center *= height; roi.width *= width; roi.height *= height;
    }
    g.roi_w = roi.width;
    g.roi_h = roi.height;
    g.warp_w = keep_aspect ? static_cast<int>(roi.width) : out_w;   // Rect::size truncates (types.rs:52-59), 203-207
    g.warp_h = keep_aspect ? static_cast<int>(roi.height) : out_h;
    if (g.warp_w <= 0 || g.warp_h <= 0) throw std::runtime_error("ROI is empty");
    // Rect::points (types.rs:80-96), cast to f32 as Point2f (transform.rs:210-213)
    const double x = roi.x_center, y = roi.y_center, hw = roi.width / 2.0, hh = roi.height / 2.0;
    double p[4][2] = {{x - hw, y - hh}, {x + hw, y - hh}, {x + hw, y + hh}, {x - hw, y + hh}};
    if (roi.rotation != 0.0) {
        const double s = std::sin(roi.rotation), c = std::cos(roi.rotation);
        for (auto& q : p) {
            const double dx = q[0] - x, dy = q[1] - y;
            q[0] = x + dx * c - dy * s;
            q[1] = y + dx * s + dy * c;
        }
    }
    float src[4][2], dst[4][2] = {{0, 0}, {static_cast<float>(g.warp_w), 0}, {static_cast<float>(g.warp_w), static_cast<float>(g.warp_h)}, {0, static_cast<float>(g.warp_h)}};
    for (int i = 0; i < 4; i++) { src[i][0] = static_cast<float>(p[i][0]); src[i][1] = static_cast<float>(p[i][1]); }
    double M[9];
    if (!homography(src, dst, M) || !inverse3(M, g.Minv)) throw std::runtime_error("degenerate ROI (singular perspective transform)");
    g.new_w = static_cast<int>(roi.width);
    g.new_h = static_cast<int>(roi.height);
    if (keep_aspect) {  // transform.rs:239-257
        const double out_aspect = static_cast<double>(out_h / out_w);  // integer division in the reference
        const double roi_aspect = roi.height / roi.width;
        if (out_aspect > roi_aspect) { g.new_h = static_cast<int>(roi.width * out_aspect); g.pad_y = (1.0 - roi_aspect / out_aspect) / 2.0; }
        else { g.new_w = static_cast<int>(roi.height / out_aspect); g.pad_x = (1.0 - out_aspect / roi_aspect) / 2.0; }
        if (g.new_w != static_cast<int>(roi.width) || g.new_h != static_cast<int>(roi.height)) {
            g.two_stage = true;
            g.pad_h = static_cast<int>(g.pad_x * g.new_w);
            g.pad_v = static_cast<int>(g.pad_y * g.new_h);
        }
    }
    return g;
}

size_t align256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

// ---------------------------------------------------------------------------------------------- kernels
struct WarpArgs {
    const uint8_t* src; int sw, sh, sstride;
    uint8_t* dst; int dw, dh;
    double M[9];
    int bw0, bh0;
};

// cv::warpPerspective, INTER_LINEAR, BORDER_CONSTANT(0), 8UC3 (generic fixed-point path).
__global__ void warp_u8_kernel(WarpArgs a) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.dw || y >= a.dh) return;
    // OpenCV evaluates the mapping per tile: X0 at the tile's first column, then + M[0]*x1
    int bx = (x / a.bw0) * a.bw0, x1 = x - bx;
    double X0 = a.M[0] * bx + a.M[1] * y + a.M[2];
    double Y0 = a.M[3] * bx + a.M[4] * y + a.M[5];
    double W = (a.M[6] * bx + a.M[7] * y + a.M[8]) + a.M[6] * x1;
    W = W != 0.0 ? 32.0 / W : 0.0;
    double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + a.M[0] * x1) * W));
    double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + a.M[3] * x1) * W));
    int X = __double2int_rn(fX), Y = __double2int_rn(fY);
    int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);
    int ax = X & 31, ay = Y & 31;
    int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
    if (ax == 0 && ay == 0) { w00 = 32767; w11 = 1; }  // saturate_cast<short>(32768) + sum fix-up of the weight table
    uint8_t* d = a.dst + ((long)y * a.dw + x) * 3;
    if (sx >= a.sw || sx + 1 < 0 || sy >= a.sh || sy + 1 < 0) { d[0] = d[1] = d[2] = 0; return; }
    bool x0in = sx >= 0 && sx < a.sw, x1in = sx + 1 >= 0 && sx + 1 < a.sw;
    bool y0in = sy >= 0 && sy < a.sh, y1in = sy + 1 >= 0 && sy + 1 < a.sh;
    const uint8_t* r0 = a.src + (long)sy * a.sstride + (long)sx * 3;
    const uint8_t* r1 = r0 + a.sstride;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int p00 = (y0in && x0in) ? r0[c] : 0, p01 = (y0in && x1in) ? r0[3 + c] : 0;
        int p10 = (y1in && x0in) ? r1[c] : 0, p11 = (y1in && x1in) ? r1[3 + c] : 0;
        int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
        d[c] = (uint8_t)min(max(v, 0), 255);
    }
}

struct ResizeArgs {
    const uint8_t* src; int sw, sh;   // un-padded source image [sh][sw][3]
    int pad_l, pad_t, pw, ph;         // virtual zero border: padded size pw x ph (copyMakeBorder fused)
    uint8_t* dst; int dw, dh;
    double scale_x, scale_y;
    int area2x;
};

__device__ __forceinline__ int padded_px(const ResizeArgs& a, int y, int x, int c) {
    int yy = y - a.pad_t, xx = x - a.pad_l;
    if (yy < 0 || yy >= a.sh || xx < 0 || xx >= a.sw) return 0;
    return a.src[((long)yy * a.sw + xx) * 3 + c];
}

// cv::resize INTER_LINEAR 8UC3: 11-bit coefficient pairs, int32 horizontal pass, (b*(S>>4))>>16 vertical pass.
__global__ void resize_u8_kernel(ResizeArgs a) {
    int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y * blockDim.y + threadIdx.y;
    if (dx >= a.dw || dy >= a.dh) return;
    uint8_t* d = a.dst + ((long)dy * a.dw + dx) * 3;
    if (a.area2x) {  // exact 2x shrink: INTER_LINEAR is routed to INTER_AREA
#pragma unroll
        for (int c = 0; c < 3; c++)
            d[c] = (uint8_t)((padded_px(a, 2 * dy, 2 * dx, c) + padded_px(a, 2 * dy, 2 * dx + 1, c) + padded_px(a, 2 * dy + 1, 2 * dx, c) +
                              padded_px(a, 2 * dy + 1, 2 * dx + 1, c) + 2) >> 2);
        return;
    }
    float fx = (float)(((double)dx + 0.5) * a.scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= a.pw - 1) { fx = 0.f; sx = a.pw - 1; }
    float fy = (float)(((double)dy + 0.5) * a.scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    int a0 = min(max(__float2int_rn((1.f - fx) * 2048.f), -32768), 32767), a1 = min(max(__float2int_rn(fx * 2048.f), -32768), 32767);
    int b0 = min(max(__float2int_rn((1.f - fy) * 2048.f), -32768), 32767), b1 = min(max(__float2int_rn(fy * 2048.f), -32768), 32767);
    int sy0 = min(max(sy, 0), a.ph - 1), sy1 = min(max(sy + 1, 0), a.ph - 1);
    bool edge = sx + 1 >= a.pw;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int r0, r1;
        if (!edge) {
            r0 = padded_px(a, sy0, sx, c) * a0 + padded_px(a, sy0, sx + 1, c) * a1;
            r1 = padded_px(a, sy1, sx, c) * a0 + padded_px(a, sy1, sx + 1, c) * a1;
        } else {
            r0 = padded_px(a, sy0, sx, c) * 2048;
            r1 = padded_px(a, sy1, sx, c) * 2048;
        }
        int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        d[c] = (uint8_t)min(max(v, 0), 255);
    }
}

// flip + `(pixel as f64 * (max - min) / 255.0 + min) as f32` (transform.rs:282-301)
__global__ void normalize_kernel(const uint8_t* src, int w, int h, int flip, double rmin, double rmax, float* out) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    int sx = flip ? w - 1 - x : x;
    const uint8_t* s = src + ((long)y * w + sx) * 3;
    float* o = out + ((long)y * w + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) o[c] = (float)((double)s[c] * (rmax - rmin) / 255.0 + rmin);
}

void launch_resize(const uint8_t* src, int sw, int sh, int pad_l, int pad_t, uint8_t* dst, int dw, int dh, hipStream_t s) {
    ResizeArgs a{};
    a.src = src; a.sw = sw; a.sh = sh; a.pad_l = pad_l; a.pad_t = pad_t; a.pw = sw + 2 * pad_l; a.ph = sh + 2 * pad_t;
    a.dst = dst; a.dw = dw; a.dh = dh;
    const double inv_x = static_cast<double>(dw) / a.pw, inv_y = static_cast<double>(dh) / a.ph;
    a.scale_x = 1.0 / inv_x;
    a.scale_y = 1.0 / inv_y;
    const int isx = static_cast<int>(std::lrint(a.scale_x)), isy = static_cast<int>(std::lrint(a.scale_y));
    a.area2x = std::fabs(a.scale_x - isx) < 2.220446049250313e-16 && std::fabs(a.scale_y - isy) < 2.220446049250313e-16 && isx == 2 && isy == 2;
    dim3 blk(32, 8), grd((dw + 31) / 32, (dh + 7) / 8);
    hipLaunchKernelGGL(resize_u8_kernel, grd, blk, 0, s, a);
    hip_check(hipGetLastError(), "resize kernel launch");
}

}  // namespace

size_t image_to_tensor_scratch_bytes(int width, int height, int stride, const mi_rect* roi, int out_w, int out_h, bool keep_aspect) {
    Geom g = make_geom(width, height, roi, out_w, out_h, keep_aspect);
    size_t n = align256(static_cast<size_t>(stride) * height) + align256(static_cast<size_t>(g.warp_w) * g.warp_h * 3);
    if (keep_aspect) n += align256(static_cast<size_t>(g.new_w) * g.new_h * 3) + align256(static_cast<size_t>(out_w) * out_h * 3);
    return n + 256;
}

void image_to_tensor_device(const uint8_t* rgb_host, int width, int height, int stride, const mi_rect* roi, int out_w, int out_h,
                            bool keep_aspect, double range_min, double range_max, bool flip, float* d_out, double padding[4],
                            void* d_scratch, hipStream_t s) {
    Geom g = make_geom(width, height, roi, out_w, out_h, keep_aspect);
    uint8_t* base = static_cast<uint8_t*>(d_scratch);
    uint8_t* d_img = base;
    uint8_t* d_warp = d_img + align256(static_cast<size_t>(stride) * height);
    uint8_t* d_mid = d_warp + align256(static_cast<size_t>(g.warp_w) * g.warp_h * 3);
    uint8_t* d_fin = d_mid + align256(static_cast<size_t>(g.new_w) * g.new_h * 3);
    hip_check(hipMemcpyAsync(d_img, rgb_host, static_cast<size_t>(stride) * height, hipMemcpyHostToDevice, s), "H2D image");

    WarpArgs wa{};
    wa.src = d_img; wa.sw = width; wa.sh = height; wa.sstride = stride;
    wa.dst = d_warp; wa.dw = g.warp_w; wa.dh = g.warp_h;
    std::memcpy(wa.M, g.Minv, sizeof wa.M);
    // OpenCV tile geometry (BLOCK_SZ = 32): bh0 = min(16, h); bw0 = min(1024 / bh0, w); bh0 = min(1024 / bw0, h)
    wa.bh0 = std::min(16, g.warp_h);
    wa.bw0 = std::min(1024 / wa.bh0, g.warp_w);
    wa.bh0 = std::min(1024 / wa.bw0, g.warp_h);
    {
        dim3 blk(32, 8), grd((g.warp_w + 31) / 32, (g.warp_h + 7) / 8);
        hipLaunchKernelGGL(warp_u8_kernel, grd, blk, 0, s, wa);
        hip_check(hipGetLastError(), "warp kernel launch");
    }
    const uint8_t* cur = d_warp;
    int cw = g.warp_w, ch = g.warp_h;
    if (keep_aspect) {
        if (g.two_stage) {
            if (cw + 2 * g.pad_h == g.new_w && ch + 2 * g.pad_v == g.new_h) {
                // same-size resize is a copy in OpenCV: fold the border into the next resize instead of materialising it
                launch_resize(cur, cw, ch, g.pad_h, g.pad_v, d_fin, out_w, out_h, s);
                cur = d_fin; cw = out_w; ch = out_h;
            } else {
                launch_resize(cur, cw, ch, g.pad_h, g.pad_v, d_mid, g.new_w, g.new_h, s);
                cur = d_mid; cw = g.new_w; ch = g.new_h;
            }
        }
        if (!(cur == d_fin)) {
            if (cw == out_w && ch == out_h) {
                // cv::resize to the same size copies
            } else {
                launch_resize(cur, cw, ch, 0, 0, d_fin, out_w, out_h, s);
                cur = d_fin; cw = out_w; ch = out_h;
            }
        }
    }
    {
        dim3 blk(32, 8), grd((cw + 31) / 32, (ch + 7) / 8);
        hipLaunchKernelGGL(normalize_kernel, grd, blk, 0, s, cur, cw, ch, flip ? 1 : 0, range_min, range_max, d_out);
        hip_check(hipGetLastError(), "normalize kernel launch");
    }
    padding[0] = g.pad_x; padding[1] = g.pad_y; padding[2] = g.pad_x; padding[3] = g.pad_y;
}

}  // namespace mi
