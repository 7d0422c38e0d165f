// preproc.hip — transform::image_to_tensor on the GPU (reference: /root/reference/src/face_detection_lite/
// transform.rs:188-309), plus the ROI maths that links the detector, the mesh and the iris stages
// (face_landmark.rs:180-198, iris_landmark.rs:268-292, transform.rs:44-109) so that a whole batch can go through
// detector -> mesh -> iris without leaving the device.
//
// The reference chains OpenCV calls on the CPU: getPerspectiveTransform, warpPerspective (INTER_LINEAR, BORDER_CONSTANT 0),
// copyMakeBorder, resize (INTER_LINEAR), flip, then a per-pixel u8 -> f32 loop.  Every stage rounds to u8, so the
// stages cannot be merged arithmetically — but they can be merged *structurally*: one thread per output pixel evaluates
// the chain backwards (output pixel <- 2x2 of the resized image <- 2x2 each of the bordered/resized image <- 2x2 each of
// the warped image <- 2x2 source pixels), recomputing the small intermediate neighbourhoods instead of materialising
// variable-sized intermediate images.  Each stage keeps OpenCV's generic 8-bit fixed-point arithmetic (1/32-pixel
// coordinates + 15-bit weights for the warp, 11-bit coefficient pairs for the resize), so results stay within 1 LSB of
// a non-IPP OpenCV build.  The geometry (homography, sizes, paddings) is computed per item by the same code on host
// (single image) or device (batches whose ROIs were produced on the device).
#include "preproc.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "engine.hpp"
#include "roi_dev.hpp"

namespace mi {
namespace {

// ---------------------------------------------------------------------------------------------- geometry (host + device)
// `ws`: the 8x9 system, element (i, j) at ws[(i * 9 + j) * wstride].  The row swaps index it dynamically, so a local array lives in
// scratch memory on the device (a dependent HBM round trip per access: 83 us per launch for one lane's elimination); the batched kernel
// passes LDS instead, lane-minor (wstride = lanes per workgroup: the 64 lanes of a wave touch 64 consecutive doubles, no bank conflicts —
// one 8x9 block per lane was a 16-way conflict on every access, 35 us per launch).
__host__ __device__ inline bool solve_homography(const float src[4][2], const float dst[4][2], double M[9], double* ws = nullptr, int wstride = 1) {
    // 8x8 system of cv::getPerspectiveTransform, Gauss-Jordan with partial pivoting in f64
    double local[72];
    double* base = ws ? ws : local;
    const int st = ws ? wstride : 1;
    auto a = [&](int i, int j) -> double& { return base[(i * 9 + j) * st]; };
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 9; j++) a(i, j) = 0.0;
    for (int i = 0; i < 4; i++) {
        const double X = src[i][0], Y = src[i][1], u = dst[i][0], v = dst[i][1];
        a(i, 0) = X; a(i, 1) = Y; a(i, 2) = 1; a(i, 6) = -X * u; a(i, 7) = -Y * u; a(i, 8) = u;
        a(i + 4, 3) = X; a(i + 4, 4) = Y; a(i + 4, 5) = 1; a(i + 4, 6) = -X * v; a(i + 4, 7) = -Y * v; a(i + 4, 8) = v;
    }
    for (int c = 0; c < 8; c++) {
        int p = c;
        for (int r = c + 1; r < 8; r++)
            if (fabs(a(r, c)) > fabs(a(p, c))) p = r;
        if (fabs(a(p, c)) < 1e-300) return false;
        if (p != c)
            for (int k = 0; k < 9; k++) { double t = a(c, k); a(c, k) = a(p, k); a(p, k) = t; }
        for (int r = 0; r < 8; r++) {
            if (r == c) continue;
            const double f = a(r, c) / a(c, c);
            if (f != 0.0)
                for (int k = c; k < 9; k++) a(r, k) -= f * a(c, k);
        }
    }
    for (int i = 0; i < 8; i++) M[i] = a(i, 8) / a(i, i);
    M[8] = 1.0;
    return true;
}

#ifdef __HIP_DEVICE_COMPILE__
// The same elimination with one ROW per lane: the 8 lanes of an aligned group (sub = lane % 8) hold the 8 x 9 system in registers,
// pivot search / row swap / pivot-row broadcast by shuffles.  Every element sees the operations of solve_homography() in the same
// order (a[r][k] -= (a[r][c] / a[c][c]) * a[c][k], separate multiply and subtract), so the result is bit-identical; one lane walking
// the system through LDS was a chain of ~1000 dependent LDS round trips (32 us per launch for a few hundred ROIs).
__device__ inline bool solve_homography_lanes(const float src[4][2], const float dst[4][2], double M[9], int sub) {
    double a[9];
#pragma unroll
    for (int j = 0; j < 9; j++) a[j] = 0.0;
    {
        const int i = sub & 3;
        const float Xf = i == 0 ? src[0][0] : i == 1 ? src[1][0] : i == 2 ? src[2][0] : src[3][0], Yf = i == 0 ? src[0][1] : i == 1 ? src[1][1] : i == 2 ? src[2][1] : src[3][1];
        const float uf = i == 0 ? dst[0][0] : i == 1 ? dst[1][0] : i == 2 ? dst[2][0] : dst[3][0], vf = i == 0 ? dst[0][1] : i == 1 ? dst[1][1] : i == 2 ? dst[2][1] : dst[3][1];
        const double X = Xf, Y = Yf, u = uf, v = vf;
        if (sub < 4) { a[0] = X; a[1] = Y; a[2] = 1; a[6] = -X * u; a[7] = -Y * u; a[8] = u; }
        else { a[3] = X; a[4] = Y; a[5] = 1; a[6] = -X * v; a[7] = -Y * v; a[8] = v; }
    }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const double mag = fabs(a[c]);
        int p = c;
        double best = __shfl(mag, c, 8);
#pragma unroll
        for (int r = c + 1; r < 8; r++) {
            const double m = __shfl(mag, r, 8);
            if (m > best) { best = m; p = r; }
        }
        if (best < 1e-300) ok = false;   // uniform over the group; the elimination goes on (its result is discarded) so that the shuffles stay convergent
        const int partner = sub == c ? p : (sub == p ? c : sub);
        double piv[9];
#pragma unroll
        for (int k = 0; k < 9; k++) a[k] = __shfl(a[k], partner, 8);
#pragma unroll
        for (int k = 0; k < 9; k++) piv[k] = __shfl(a[k], c, 8);
        if (sub != c) {
            const double f = a[c] / piv[c];
            if (f != 0.0) {
#pragma unroll
                for (int k = c; k < 9; k++) a[k] -= f * piv[k];
            }
        }
    }
    double diag = a[0];
#pragma unroll
    for (int j = 1; j < 8; j++) diag = sub == j ? a[j] : diag;
    const double mi = a[8] / diag;
#pragma unroll
    for (int j = 0; j < 8; j++) M[j] = __shfl(mi, j, 8);
    M[8] = 1.0;
    return ok;
}
#endif

__host__ __device__ inline bool inverse3(const double m[9], double o[9]) {
    const double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    o[0] = c0 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c1 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c2 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

// transform.rs:190-257 — everything that depends only on (image size, ROI, output size, keep_aspect_ratio)
// lanes_sub >= 0 (device only): the 8 lanes of an aligned group compute the SAME item together (solve_homography_lanes), lanes_sub = lane % 8
__host__ __device__ inline PreGeom compute_geom(int width, int height, const RectD* roi_in, int out_w, int out_h, bool keep_aspect, double* ws = nullptr, int wstride = 1, int lanes_sub = -1) {
    PreGeom g;
    for (int i = 0; i < 9; i++) g.Minv[i] = 0.0;
    g.pad_x = g.pad_y = 0.0;
    g.stage1 = 0; g.pad_h = g.pad_v = 0; g.stage2 = 0; g.valid = 0; g.bw0 = 1;
    double xc = 0.5, yc = 0.5, rw = 1.0, rh = 1.0, rot = 0.0;  // default ROI = whole image, normalised (190-197)
    bool normalized = true;
    if (roi_in) { xc = roi_in->x_center; yc = roi_in->y_center; rw = roi_in->width; rh = roi_in->height; rot = roi_in->rotation; normalized = roi_in->normalized != 0; }
    if (normalized) { xc *= width; yc *= height; rw *= width; rh *= height; }  // Rect::scaled(size, false), types.rs:62-77
    g.warp_w = keep_aspect ? (int)rw : out_w;   // Rect::size truncates (types.rs:52-59), transform.rs:203-207
    g.warp_h = keep_aspect ? (int)rh : out_h;
    g.new_w = (int)rw;
    g.new_h = (int)rh;
    if (!(g.warp_w > 0 && g.warp_h > 0 && g.warp_w <= 16384 && g.warp_h <= 16384)) return g;
    // Rect::points (types.rs:80-96) -> Point2f (transform.rs:210-213)
    const double hw = rw / 2.0, hh = rh / 2.0;
    double p[4][2] = {{xc - hw, yc - hh}, {xc + hw, yc - hh}, {xc + hw, yc + hh}, {xc - hw, yc + hh}};
    if (rot != 0.0) {
        const double s = sin(rot), c = cos(rot);
        for (int i = 0; i < 4; i++) {
            const double dx = p[i][0] - xc, dy = p[i][1] - yc;
            p[i][0] = xc + dx * c - dy * s;
            p[i][1] = yc + dx * s + dy * c;
        }
    }
    float src[4][2], dst[4][2] = {{0.f, 0.f}, {(float)g.warp_w, 0.f}, {(float)g.warp_w, (float)g.warp_h}, {0.f, (float)g.warp_h}};
    for (int i = 0; i < 4; i++) { src[i][0] = (float)p[i][0]; src[i][1] = (float)p[i][1]; }
    double M[9];
    bool solved;
#ifdef __HIP_DEVICE_COMPILE__
    if (lanes_sub >= 0) solved = solve_homography_lanes(src, dst, M, lanes_sub);
    else
#endif
        solved = solve_homography(src, dst, M, ws, wstride);
    (void)lanes_sub;
    if (!solved || !inverse3(M, g.Minv)) return g;
    // OpenCV tile geometry of warpPerspective (BLOCK_SZ = 32): bh0 = min(16, h); bw0 = min(1024 / bh0, w)
    const int bh0 = g.warp_h < 16 ? g.warp_h : 16;
    g.bw0 = (1024 / bh0) < g.warp_w ? (1024 / bh0) : g.warp_w;
    int cw = g.warp_w, ch = g.warp_h;
    if (keep_aspect) {  // transform.rs:239-280
        const double out_aspect = (double)(out_h / out_w);  // integer division in the reference
        const double roi_aspect = rh / rw;
        if (out_aspect > roi_aspect) { g.new_h = (int)(rw * out_aspect); g.pad_y = (1.0 - roi_aspect / out_aspect) / 2.0; }
        else { g.new_w = (int)(rh / out_aspect); g.pad_x = (1.0 - out_aspect / roi_aspect) / 2.0; }
        if (g.new_w != (int)rw || g.new_h != (int)rh) {
            g.pad_h = (int)(g.pad_x * (double)g.new_w);
            g.pad_v = (int)(g.pad_y * (double)g.new_h);
            if (!(g.new_w > 0 && g.new_h > 0 && g.new_w <= 16384 && g.new_h <= 16384)) return g;
            g.stage1 = (cw + 2 * g.pad_h == g.new_w && ch + 2 * g.pad_v == g.new_h) ? 1 : 2;  // cv::resize to the same size copies
            cw = g.new_w; ch = g.new_h;
        }
        g.stage2 = (cw == out_w && ch == out_h) ? 0 : 1;
    }
    g.valid = 1;
    return g;
}

// ---------------------------------------------------------------------------------------------- pixel chain (device)
struct Px { int r, g, b; };

// cv::warpPerspective, INTER_LINEAR, BORDER_CONSTANT(0), 8UC3 (generic fixed-point path), pixel (x, y) of the warped image
__device__ __forceinline__ Px warp_px(const PreGeom& g, const uint8_t* src, int sw, int sh, int sstride, int y, int x) {
    // OpenCV evaluates the mapping per tile: X0 at the tile's first column, then + M[0]*x1
    const int bx = (x / g.bw0) * g.bw0, x1 = x - bx;
    const double X0 = g.Minv[0] * bx + g.Minv[1] * y + g.Minv[2];
    const double Y0 = g.Minv[3] * bx + g.Minv[4] * y + g.Minv[5];
    double W = (g.Minv[6] * bx + g.Minv[7] * y + g.Minv[8]) + g.Minv[6] * x1;
    W = W != 0.0 ? 32.0 / W : 0.0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + g.Minv[0] * x1) * W));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + g.Minv[3] * x1) * W));
    const int X = __double2int_rn(fX), Y = __double2int_rn(fY);
    const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);
    const int ax = X & 31, ay = Y & 31;
    int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
    if (ax == 0 && ay == 0) { w00 = 32767; w11 = 1; }  // saturate_cast<short>(32768) + sum fix-up of the weight table
    Px o = {0, 0, 0};
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) return o;
    const bool x0in = sx >= 0 && sx < sw, x1in = sx + 1 >= 0 && sx + 1 < sw;
    const bool y0in = sy >= 0 && sy < sh, y1in = sy + 1 >= 0 && sy + 1 < sh;
    const uint8_t* r0 = src + (long)sy * sstride + (long)sx * 3;
    const uint8_t* r1 = r0 + sstride;
    int v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int p00 = (y0in && x0in) ? r0[c] : 0, p01 = (y0in && x1in) ? r0[3 + c] : 0;
        const int p10 = (y1in && x0in) ? r1[c] : 0, p11 = (y1in && x1in) ? r1[3 + c] : 0;
        v[c] = min(max((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15, 0), 255);
    }
    o.r = v[0]; o.g = v[1]; o.b = v[2];
    return o;
}

// cv::resize INTER_LINEAR 8UC3 (11-bit coefficient pairs, int32 horizontal pass, (b*(S>>4))>>16 vertical pass; exact 2x
// shrink is routed to INTER_AREA) of a virtual image `fetch(y, x)` of size pw x ph.
template <typename Fetch>
__device__ __forceinline__ Px resize_px(int pw, int ph, int dw, int dh, int dy, int dx, Fetch fetch) {
    const double scale_x = 1.0 / ((double)dw / pw), scale_y = 1.0 / ((double)dh / ph);
    const int isx = __double2int_rn(scale_x), isy = __double2int_rn(scale_y);
    Px o;
    if (fabs(scale_x - isx) < 2.220446049250313e-16 && fabs(scale_y - isy) < 2.220446049250313e-16 && isx == 2 && isy == 2) {
        const Px a = fetch(2 * dy, 2 * dx), b = fetch(2 * dy, 2 * dx + 1), c = fetch(2 * dy + 1, 2 * dx), d = fetch(2 * dy + 1, 2 * dx + 1);
        o.r = (a.r + b.r + c.r + d.r + 2) >> 2; o.g = (a.g + b.g + c.g + d.g + 2) >> 2; o.b = (a.b + b.b + c.b + d.b + 2) >> 2;
        return o;
    }
    float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= pw - 1) { fx = 0.f; sx = pw - 1; }
    float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int a0 = min(max(__float2int_rn((1.f - fx) * 2048.f), -32768), 32767), a1 = min(max(__float2int_rn(fx * 2048.f), -32768), 32767);
    const int b0 = min(max(__float2int_rn((1.f - fy) * 2048.f), -32768), 32767), b1 = min(max(__float2int_rn(fy * 2048.f), -32768), 32767);
    const int sy0 = min(max(sy, 0), ph - 1), sy1 = min(max(sy + 1, 0), ph - 1);
    const bool edge = sx + 1 >= pw;
    const Px p00 = fetch(sy0, sx), p10 = fetch(sy1, sx);
    Px p01 = p00, p11 = p10;
    if (!edge) { p01 = fetch(sy0, sx + 1); p11 = fetch(sy1, sx + 1); }
    auto mix = [&](int v00, int v01, int v10, int v11) {
        const int r0 = edge ? v00 * 2048 : v00 * a0 + v01 * a1;
        const int r1 = edge ? v10 * 2048 : v10 * a0 + v11 * a1;
        return min(max((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2, 0), 255);
    };
    o.r = mix(p00.r, p01.r, p10.r, p11.r);
    o.g = mix(p00.g, p01.g, p10.g, p11.g);
    o.b = mix(p00.b, p01.b, p10.b, p11.b);
    return o;
}

__global__ __launch_bounds__(64) void pre_geom_kernel(PreItems it, PreGeom* geom, double* padding) {
    // eight lanes per item (the 8 x 8 system of the homography: one row per lane), eight items per workgroup
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, sub = threadIdx.x & 7;
    if (i >= it.N) return;   // whole groups of 8 lanes
    PreGeom g;
    if (it.roi_valid && !it.roi_valid[i]) {
        g = compute_geom(it.width, it.height, nullptr, it.out_w, it.out_h, it.keep_aspect != 0, nullptr, 1, sub);
        g.valid = 0;
        g.pad_x = g.pad_y = 0.0;
    } else {
        g = compute_geom(it.width, it.height, it.rois ? &it.rois[i] : nullptr, it.out_w, it.out_h, it.keep_aspect != 0, nullptr, 1, sub);
    }
    if (sub) return;
    geom[i] = g;
    if (padding) { padding[4 * i] = g.pad_x; padding[4 * i + 1] = g.pad_y; padding[4 * i + 2] = g.pad_x; padding[4 * i + 3] = g.pad_y; }
}

// One thread per output pixel: flip + `(pixel as f64 * (max - min) / 255.0 + min) as f32` (transform.rs:282-301) of the
// image produced by warp -> [border (+ resize)] -> [resize].
__device__ __forceinline__ void pre_tensor_px(const PreItems& it, const PreGeom& g, int i, bool flip, int idx, float* __restrict__ out) {
    const int y = idx / it.out_w, x = idx - y * it.out_w;
    float* o = out + ((long)i * it.out_h * it.out_w + idx) * 3;
    if (!g.valid) { o[0] = o[1] = o[2] = 0.f; return; }
    const uint8_t* src = it.frames + (long)(i / it.items_per_frame) * it.frame_bytes;
    const int sx_out = flip ? it.out_w - 1 - x : x;   // cv::flip(…, 1) of the final image
    const int sw = it.width, sh = it.height, ss = it.stride;
    auto warped = [&](int yy, int xx) { return warp_px(g, src, sw, sh, ss, yy, xx); };
    auto bordered = [&](int yy, int xx) {  // copyMakeBorder(top = bottom = pad_v, left = right = pad_h, 0)
        const int wy = yy - g.pad_v, wx = xx - g.pad_h;
        if (wy < 0 || wy >= g.warp_h || wx < 0 || wx >= g.warp_w) { Px z = {0, 0, 0}; return z; }
        return warped(wy, wx);
    };
    auto stage1 = [&](int yy, int xx) {
        if (g.stage1 == 0) return warped(yy, xx);
        if (g.stage1 == 1) return bordered(yy, xx);
        return resize_px(g.warp_w + 2 * g.pad_h, g.warp_h + 2 * g.pad_v, g.new_w, g.new_h, yy, xx, bordered);
    };
    Px p;
    if (g.stage2) {
        const int cw = g.stage1 ? g.new_w : g.warp_w, ch = g.stage1 ? g.new_h : g.warp_h;
        p = resize_px(cw, ch, it.out_w, it.out_h, y, sx_out, stage1);
    } else {
        p = stage1(y, sx_out);
    }
    const double k = it.range_max - it.range_min;
    o[0] = (float)((double)p.r * k / 255.0 + it.range_min);
    o[1] = (float)((double)p.g * k / 255.0 + it.range_min);
    o[2] = (float)((double)p.b * k / 255.0 + it.range_min);
}

__global__ __launch_bounds__(256) void pre_tensor_kernel(PreItems it, const PreGeom* __restrict__ geom, float* __restrict__ out) {
    const int i = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= it.out_w * it.out_h) return;
    const PreGeom g = geom[i];
    pre_tensor_px(it, g, i, it.flip && it.flip[i], idx, out);
}

// One picture, its geometry and flip flag as kernel arguments (the single-image entries: nothing to upload but the picture)
__global__ __launch_bounds__(256) void pre_tensor_one_kernel(PreItems it, PreGeom g, int flip, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= it.out_w * it.out_h) return;
    pre_tensor_px(it, g, 0, flip != 0, idx, out);
}

// ---------------------------------------------------------------------------------------------- ROI maths (device)
__global__ void face_roi_kernel(const float* dets, const int* counts, int B, int cap, int image_w, int image_h, RectD* rois, int* valid) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    RectD r;
    valid[b] = face_roi_dev(dets + (long)b * cap * 17, counts[b], image_w, image_h, &r);   // roi_dev.hpp
    rois[b] = r;
}

__global__ void iris_roi_kernel(const float* lm, const int* present, int B, int image_w, int image_h, RectD* rois, int* valid, int* flip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * B) return;
    const int b = i >> 1, eye = i & 1;  // 0 = left (33,133), 1 = right (362,263) — iris_landmark.rs:29-35
    RectD r = {0.5, 0.5, 1.0, 1.0, 0.0, 1, 0};
    int ok = 0;
    if (present[b]) {
        const float* p0 = lm + ((long)b * 468 + (eye ? 362 : 33)) * 3;
        const float* p1 = lm + ((long)b * 468 + (eye ? 263 : 133)) * 3;
        const double ax = p0[0], ay = p0[1], bx = p1[0], by = p1[1];
        const double bbox[4] = {fmin(ax, bx), fmin(ay, by), fmax(ax, bx), fmax(ay, by)};  // bbox_from_landmarks, transform.rs:146-165
        const double kp[4] = {ax, ay, bx, by};
        ok = bbox_to_roi_dev(bbox, image_w, image_h, kp, 2.3, &r) ? 1 : 0;  // ROI_SCALE (iris_landmark.rs:27)
    }
    rois[i] = r;
    valid[i] = ok;
    flip[i] = eye;  // right eye is flipped (lib.rs:39)
}

size_t align256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

}  // namespace

void launch_pre_geom(const PreItems& it, PreGeom* d_geom, double* d_padding, hipStream_t s) {
    if (it.N <= 0) return;
    hipLaunchKernelGGL(pre_geom_kernel, dim3((it.N + 7) / 8), dim3(64), 0, s, it, d_geom, d_padding);
    hip_check(hipGetLastError(), "pre_geom kernel launch");
}

// The geometry of N whole pictures of one size (no ROI: the detector's letterbox, face_detection.rs:226) does not depend on the call: computed
// on the host by the same code and uploaded — the caller keeps it for as long as (width, height, N) stay what they are.  Synchronous.
void upload_whole_image_geom(int width, int height, int out_w, int out_h, bool keep_aspect, int N, PreGeom* d_geom, double* d_padding, hipStream_t s) {
    if (N <= 0) return;
    const PreGeom g = compute_geom(width, height, nullptr, out_w, out_h, keep_aspect);
    std::vector<PreGeom> hg(static_cast<size_t>(N), g);
    std::vector<double> hp(4 * static_cast<size_t>(N));
    for (int i = 0; i < N; i++) { hp[4 * i] = g.pad_x; hp[4 * i + 1] = g.pad_y; hp[4 * i + 2] = g.pad_x; hp[4 * i + 3] = g.pad_y; }
    hip_check(hipMemcpyAsync(d_geom, hg.data(), hg.size() * sizeof(PreGeom), hipMemcpyHostToDevice, s), "H2D geometry");
    if (d_padding) hip_check(hipMemcpyAsync(d_padding, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice, s), "H2D padding");
    hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
}

void launch_pre_tensor(const PreItems& it, const PreGeom* d_geom, float* d_out, hipStream_t s) {
    if (it.N <= 0) return;
    dim3 grid((it.out_w * it.out_h + 255) / 256, it.N);
    hipLaunchKernelGGL(pre_tensor_kernel, grid, dim3(256), 0, s, it, d_geom, d_out);
    hip_check(hipGetLastError(), "pre_tensor kernel launch");
}

void launch_face_rois(const float* d_dets, const int* d_counts, int B, int cap, int image_w, int image_h, RectD* d_rois, int* d_valid, hipStream_t s) {
    hipLaunchKernelGGL(face_roi_kernel, dim3((B + 63) / 64), dim3(64), 0, s, d_dets, d_counts, B, cap, image_w, image_h, d_rois, d_valid);
    hip_check(hipGetLastError(), "face_roi kernel launch");
}

void launch_iris_rois(const float* d_landmarks, const int* d_present, int B, int image_w, int image_h, RectD* d_rois, int* d_valid, int* d_flip,
                      hipStream_t s) {
    hipLaunchKernelGGL(iris_roi_kernel, dim3((2 * B + 63) / 64), dim3(64), 0, s, d_landmarks, d_present, B, image_w, image_h, d_rois, d_valid, d_flip);
    hip_check(hipGetLastError(), "iris_roi kernel launch");
}

size_t image_to_tensor_scratch_bytes(int width, int height, int stride, const mi_rect*, int, int, bool) {
    (void)width;
    return align256(static_cast<size_t>(stride) * height) + align256(sizeof(PreGeom)) + 256;
}

// The source rows a single picture's pre-processing can read.  With a ROI only the rows the warp samples travel to the device (an eye's ROI
// spans a tenth of the picture's rows: IrisLandmark::infer, iris_landmark.rs:188, copies 60 - 100 KB of man.jpg's 583): dst -> src is a
// homography whose denominator is positive over the target rectangle, so the source row is extreme at one of its corners; the second
// bilinear tap and OpenCV's 1/32-pixel rounding of the coordinates stay inside a margin of four rows.
static void sampled_rows(const PreGeom& g, bool has_roi, int height, int* y_first, int* y_last) {
    *y_first = 0;
    *y_last = height - 1;
    if (!has_roi || g.warp_w <= 0 || g.warp_h <= 0) return;
    double lo = 1e300, hi = -1e300;
    for (int c = 0; c < 4; c++) {
        const double x = (c & 1) ? g.warp_w : -1.0, y = (c & 2) ? g.warp_h : -1.0;
        const double w = g.Minv[6] * x + g.Minv[7] * y + g.Minv[8];
        const double sy = (g.Minv[3] * x + g.Minv[4] * y + g.Minv[5]) / w;
        if (!(w > 1e-9) || !(sy > -1e9 && sy < 1e9)) return;   // (not a map this argument covers: the whole picture)
        lo = std::min(lo, sy);
        hi = std::max(hi, sy);
    }
    *y_first = static_cast<int>(std::max(0.0, std::min(static_cast<double>(height - 1), std::floor(lo) - 4.0)));
    *y_last = static_cast<int>(std::max(static_cast<double>(*y_first), std::min(static_cast<double>(height - 1), std::ceil(hi) + 4.0)));
}

void image_to_tensor_device(const uint8_t* rgb_host, int width, int height, int stride, const mi_rect* roi, int out_w, int out_h,
                            bool keep_aspect, double range_min, double range_max, bool flip, float* d_out, double padding[4],
                            void* d_scratch, hipStream_t s) {
    static_assert(sizeof(mi_rect) == sizeof(RectD), "mi_rect layout");
    RectD r;
    if (roi) std::memcpy(&r, roi, sizeof r);
    const PreGeom g = compute_geom(width, height, roi ? &r : nullptr, out_w, out_h, keep_aspect);  // same code as the device path
    if (!g.valid) throw std::runtime_error("ROI is empty or degenerate (singular perspective transform)");
    uint8_t* d_img = static_cast<uint8_t*>(d_scratch);
    PreGeom* d_geom = reinterpret_cast<PreGeom*>(d_img + align256(static_cast<size_t>(stride) * height));
    // the caller's last row owns 3 * width bytes, not a whole stride (a cv::Mat ROI view ends there)
    int y_first = 0, y_last = height - 1;
    sampled_rows(g, roi != nullptr, height, &y_first, &y_last);
    const size_t off = static_cast<size_t>(stride) * y_first;
    const size_t bytes = static_cast<size_t>(stride) * (y_last - y_first) + (y_last == height - 1 ? static_cast<size_t>(3) * width : static_cast<size_t>(stride));
    hip_check(hipMemcpyAsync(d_img + off, rgb_host + off, bytes, hipMemcpyHostToDevice, s), "H2D image");
    hip_check(hipMemcpyAsync(d_geom, &g, sizeof g, hipMemcpyHostToDevice, s), "H2D geometry");
    PreItems it{};
    it.frames = d_img; it.frame_bytes = 0; it.width = width; it.height = height; it.stride = stride;
    it.items_per_frame = 1; it.N = 1; it.out_w = out_w; it.out_h = out_h; it.keep_aspect = keep_aspect;
    it.range_min = range_min; it.range_max = range_max;
    int* d_flip = nullptr;
    if (flip) {  // flip flag lives right after the geometry
        d_flip = reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(d_geom) + sizeof(PreGeom));
        const int one = 1;
        hip_check(hipMemcpyAsync(d_flip, &one, sizeof one, hipMemcpyHostToDevice, s), "H2D flip");
    }
    it.flip = d_flip;
    launch_pre_tensor(it, d_geom, d_out, s);
    // the H2D copies above read host stack variables: make sure they are consumed before returning
    hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    padding[0] = g.pad_x; padding[1] = g.pad_y; padding[2] = g.pad_x; padding[3] = g.pad_y;
}

void image_to_tensor_enqueue(const uint8_t* rgb_host, int width, int height, int stride, const mi_rect* roi, int out_w, int out_h,
                             bool keep_aspect, double range_min, double range_max, bool flip, float* d_out, double padding[4],
                             uint8_t* d_img, hipStream_t s) {
    RectD r;
    if (roi) std::memcpy(&r, roi, sizeof r);
    const PreGeom g = compute_geom(width, height, roi ? &r : nullptr, out_w, out_h, keep_aspect);  // same code as the device path
    if (!g.valid) throw std::runtime_error("ROI is empty or degenerate (singular perspective transform)");
    // the caller's last row owns 3 * width bytes, not a whole stride (a cv::Mat ROI view ends there)
    int y_first = 0, y_last = height - 1;
    sampled_rows(g, roi != nullptr, height, &y_first, &y_last);
    const size_t off = static_cast<size_t>(stride) * y_first;
    const size_t bytes = static_cast<size_t>(stride) * (y_last - y_first) + (y_last == height - 1 ? static_cast<size_t>(3) * width : static_cast<size_t>(stride));
    hip_check(hipMemcpyAsync(d_img + off, rgb_host + off, bytes, hipMemcpyHostToDevice, s), "H2D image");
    PreItems it{};
    it.frames = d_img; it.frame_bytes = 0; it.width = width; it.height = height; it.stride = stride;
    it.items_per_frame = 1; it.N = 1; it.out_w = out_w; it.out_h = out_h; it.keep_aspect = keep_aspect;
    it.range_min = range_min; it.range_max = range_max;
    hipLaunchKernelGGL(pre_tensor_one_kernel, dim3((out_w * out_h + 255) / 256), dim3(256), 0, s, it, g, flip ? 1 : 0, d_out);
    hip_check(hipGetLastError(), "pre_tensor kernel launch");
    padding[0] = g.pad_x; padding[1] = g.pad_y; padding[2] = g.pad_x; padding[3] = g.pad_y;
}

void image_to_tensor_enqueue_device(const uint8_t* d_img, int width, int height, int stride, const mi_rect* roi, int out_w, int out_h,
                                    bool keep_aspect, double range_min, double range_max, bool flip, float* d_out, double padding[4], hipStream_t s) {
    RectD r;
    if (roi) std::memcpy(&r, roi, sizeof r);
    const PreGeom g = compute_geom(width, height, roi ? &r : nullptr, out_w, out_h, keep_aspect);
    if (!g.valid) throw std::runtime_error("ROI is empty or degenerate (singular perspective transform)");
    PreItems it{};
    it.frames = d_img; it.frame_bytes = 0; it.width = width; it.height = height; it.stride = stride;
    it.items_per_frame = 1; it.N = 1; it.out_w = out_w; it.out_h = out_h; it.keep_aspect = keep_aspect;
    it.range_min = range_min; it.range_max = range_max;
    hipLaunchKernelGGL(pre_tensor_one_kernel, dim3((out_w * out_h + 255) / 256), dim3(256), 0, s, it, g, flip ? 1 : 0, d_out);
    hip_check(hipGetLastError(), "pre_tensor kernel launch");
    padding[0] = g.pad_x; padding[1] = g.pad_y; padding[2] = g.pad_x; padding[3] = g.pad_y;
}

}  // namespace mi
