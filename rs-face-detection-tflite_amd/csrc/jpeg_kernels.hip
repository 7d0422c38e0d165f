// jpeg_kernels.hip — device half of the JPEG path (jpeg.hpp): the sample arithmetic of libjpeg-turbo's default decoder
// (what cv::imdecode runs behind /root/reference/src/face_detection_lite/utils.rs:13), bit for bit:
//   jpeg_idct_kernel  : dequantise + jpeg_idct_islow (jidctint.c: 13-bit constants, 2 extra bits after the column pass,
//                       range limit = 10-bit wrap, +128, clamp).  8 lanes per 8x8 block: lane = column for the first
//                       pass, lane = row for the second, the 8x8 workspace transposed through LDS.
//   jpeg_color_kernel : h2v2 / h2v1 "fancy" (triangle filter) chroma upsampling (jdsample.c, context rows replicated at
//                       the picture edges) fused with the 16-bit fixed-point YCbCr -> RGB of jdcolor.c; one lane per pixel,
//                       RGB u8 written interleaved.
// Integer work, HBM-bound in principle (2 B of coefficients in, 1 B of samples out per sample; 1.5 B in, 3 B out per pixel)
// and tiny in practice (a 540x360 picture is 4590 blocks): the point is that the picture lands in HBM ready for
// image_to_tensor without a host round trip of the pixels.
#include <hip/hip_runtime.h>

#include "jpeg.hpp"

namespace mi {
namespace {

struct IdctComp { long coef_off; long plane_off; int bw, nblocks, tq, first; };
struct IdctArgs { IdctComp c[3]; int ncomp, total; };

__device__ __forceinline__ void idct_1d(const long long (&in)[8], long long (&out)[8], int shift) {
    const long long z2 = in[2], z3 = in[6];
    const long long z1 = (z2 + z3) * 4433;
    const long long tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
    const long long tmp0 = (in[0] + in[4]) << 13, tmp1 = (in[0] - in[4]) << 13;
    const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    long long t0 = in[7], t1 = in[5], t2 = in[3], t3 = in[1];
    long long z1b = t0 + t3, z2b = t1 + t2, z3b = t0 + t2, z4b = t1 + t3;
    const long long z5 = (z3b + z4b) * 9633;
    t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
    z1b *= -7373; z2b *= -20995; z3b *= -16069; z4b *= -3196;
    z3b += z5; z4b += z5;
    t0 += z1b + z3b; t1 += z2b + z4b; t2 += z2b + z3b; t3 += z1b + z4b;
    const long long rnd = 1LL << (shift - 1);
    out[0] = (tmp10 + t3 + rnd) >> shift; out[7] = (tmp10 - t3 + rnd) >> shift;
    out[1] = (tmp11 + t2 + rnd) >> shift; out[6] = (tmp11 - t2 + rnd) >> shift;
    out[2] = (tmp12 + t1 + rnd) >> shift; out[5] = (tmp12 - t1 + rnd) >> shift;
    out[3] = (tmp13 + t0 + rnd) >> shift; out[4] = (tmp13 - t0 + rnd) >> shift;
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(IdctArgs a, const int16_t* __restrict__ coef, const uint16_t* __restrict__ qt, uint8_t* __restrict__ planes) {
    __shared__ int ws[32][8][9];  // [block][row][col], padded
    const int lb = threadIdx.x >> 3, k = threadIdx.x & 7;
    const int gb = blockIdx.x * 32 + lb;
    const bool live = gb < a.total;
    int ci = 0;
    if (live) while (ci + 1 < a.ncomp && gb >= a.c[ci + 1].first) ci++;
    const IdctComp cp = a.c[ci];
    const int b = live ? gb - cp.first : 0;
    if (live) {  // column k
        const int16_t* src = coef + cp.coef_off + (long)b * 64;
        const uint16_t* q = qt + cp.tq * 64;
        long long in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; r++) in[r] = (long long)src[r * 8 + k] * (long long)q[r * 8 + k];
        idct_1d(in, out, 13 - 2);
#pragma unroll
        for (int r = 0; r < 8; r++) ws[lb][r][k] = (int)out[r];
    }
    __syncthreads();
    if (live) {  // row k
        long long in[8], out[8];
#pragma unroll
        for (int c = 0; c < 8; c++) in[c] = ws[lb][k][c];
        idct_1d(in, out, 13 + 2 + 3);
        const int by = b / cp.bw, bx = b - by * cp.bw;
        uint8_t* dst = planes + cp.plane_off + ((long)(by * 8 + k) * cp.bw + bx) * 8;
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            long long x = ((out[c] + 512) & 1023) - 512 + 128;  // range_limit[x & RANGE_MASK]
            unsigned v = (unsigned)(x < 0 ? 0 : (x > 255 ? 255 : x));
            if (c < 4) lo |= v << (8 * c); else hi |= v << (8 * (c - 4));
        }
        *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
    }
}

struct ColorArgs {
    long off[3];
    int stride[3], dw, dh;  // chroma real size
    int W, H, ncomp, hs, vs;
};

__device__ __forceinline__ int fancy_chroma(const uint8_t* pl, int st, int dw, int dh, int hs, int vs, int x, int y) {
    if (hs == 1) return pl[(long)y * st + x];
    const int cx = x >> 1;
    if (vs == 1) {  // h2v1_fancy_upsample
        const int t = pl[(long)y * st + cx];
        if (dw <= 2) return t;
        if ((x & 1) == 0) return cx == 0 ? t : (3 * t + pl[(long)y * st + cx - 1] + 1) >> 2;
        return cx == dw - 1 ? t : (3 * t + pl[(long)y * st + cx + 1] + 2) >> 2;
    }
    const int cy = y >> 1;  // h2v2_fancy_upsample
    if (dw <= 2) return pl[(long)cy * st + cx];
    int fy = (y & 1) ? cy + 1 : cy - 1;
    fy = fy < 0 ? 0 : (fy > dh - 1 ? dh - 1 : fy);
    const uint8_t *n = pl + (long)cy * st, *f = pl + (long)fy * st;
    const int tc = 3 * n[cx] + f[cx];
    if ((x & 1) == 0) return cx == 0 ? (tc * 4 + 8) >> 4 : (3 * tc + 3 * n[cx - 1] + f[cx - 1] + 8) >> 4;
    return cx == dw - 1 ? (tc * 4 + 7) >> 4 : (3 * tc + 3 * n[cx + 1] + f[cx + 1] + 7) >> 4;
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(ColorArgs a, const uint8_t* __restrict__ planes, uint8_t* __restrict__ rgb) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const int Y = planes[a.off[0] + (long)y * a.stride[0] + x];
    uint8_t* o = rgb + ((long)y * a.W + x) * 3;
    if (a.ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; return; }
    const int cb = fancy_chroma(planes + a.off[1], a.stride[1], a.dw, a.dh, a.hs, a.vs, x, y) - 128;
    const int cr = fancy_chroma(planes + a.off[2], a.stride[2], a.dw, a.dh, a.hs, a.vs, x, y) - 128;
    // jdcolor.c, SCALEBITS 16: FIX(1.40200) 91881, FIX(1.77200) 116130, FIX(0.71414) 46802, FIX(0.34414) 22554
    const int r = Y + ((91881 * cr + 32768) >> 16);
    const int g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    const int b = Y + ((116130 * cb + 32768) >> 16);
    o[0] = (uint8_t)min(max(r, 0), 255);
    o[1] = (uint8_t)min(max(g, 0), 255);
    o[2] = (uint8_t)min(max(b, 0), 255);
}

}  // namespace

size_t jpeg_plane_bytes(const JpegFrame& f) {
    size_t t = 0;
    for (int c = 0; c < f.ncomp; c++) t += static_cast<size_t>(f.comp[c].bw) * f.comp[c].bh * 64;
    return t;
}

int launch_jpeg_idct(const JpegFrame& f, const int16_t* d_coef, const uint16_t* d_qt, uint8_t* d_planes, hipStream_t s) {
    IdctArgs a{};
    a.ncomp = f.ncomp;
    long plane = 0;
    int first = 0;
    for (int c = 0; c < f.ncomp; c++) {
        const JpegComponent& cp = f.comp[c];
        a.c[c] = IdctComp{static_cast<long>(cp.coef_off), plane, cp.bw, cp.bw * cp.bh, cp.tq, first};
        plane += static_cast<long>(cp.bw) * cp.bh * 64;
        first += cp.bw * cp.bh;
    }
    a.total = first;
    if (a.total <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((a.total + 31) / 32)), dim3(256), 0, s, a, d_coef, d_qt, d_planes);
    return (int)hipGetLastError();
}

int launch_jpeg_color(const JpegFrame& f, const uint8_t* d_planes, uint8_t* d_rgb, hipStream_t s) {
    ColorArgs a{};
    long plane = 0;
    for (int c = 0; c < f.ncomp; c++) {
        a.off[c] = plane;
        a.stride[c] = f.comp[c].bw * 8;
        plane += static_cast<long>(f.comp[c].bw) * f.comp[c].bh * 64;
    }
    a.W = f.width; a.H = f.height; a.ncomp = f.ncomp; a.hs = f.hmax; a.vs = f.vmax;
    a.dw = f.ncomp == 3 ? f.comp[1].dw : f.width;
    a.dh = f.ncomp == 3 ? f.comp[1].dh : f.height;
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((f.width + 63) / 64), (unsigned)((f.height + 3) / 4)), dim3(256), 0, s, a, d_planes, d_rgb);
    return (int)hipGetLastError();
}

}  // namespace mi
