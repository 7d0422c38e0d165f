// kernels.hip — hand-written HIP kernels for gfx950 (MI355X, CDNA4; 64-wide wavefronts).
//
// Network side (replaces the TensorFlow-Lite CPU kernels behind `interpreter.invoke()`,
// /root/reference/src/face_detection_lite/face_detection.rs:235, face_landmark.rs:265, iris_landmark.rs:203):
//   conv_generic_kernel   k x k convolution (stem 5x5/3x3 s2, 2x2 s2, tiny-spatial heads) + fused epilogue
//   dw3x3_kernel          depthwise 3x3 + bias (+activation)                       (un-fused plans only)
//   block kernels         see block_kernels.hip (fused DW3x3 -> MFMA 1x1 -> skip -> activation)
//   element-wise fallbacks for graphs the fuser does not recognise
// Host-glue side (replaces Rust code of the reference crate itself):
//   ssd_postprocess_kernel  face_detection.rs:269-362 + nms.rs:56-144 + transform.rs:115-142, one workgroup/frame
//   project_landmarks_kernel transform.rs:351-432
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <algorithm>
#include <type_traits>
#include <utility>

#include "kernels.hpp"
#include "launch.hpp"
#include "roi_dev.hpp"

namespace mi {

// ------------------------------------------------------------------------------------------------ epilogue
__device__ __forceinline__ float apply_act(float v, int act, const float* __restrict__ alpha, int c) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_RELU6: return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
        case ACT_PRELU: return v >= 0.f ? v : alpha[c] * v;
        default: return v;
    }
}

// skip-connection value for output pixel (b, oy, ox), channel c (see ResMode in kernels.hpp)
__device__ __forceinline__ float load_res(const Epilogue& ep, int b, int oy, int ox, int Wo, int c) {
    if (ep.res_mode == RES_NONE || c >= ep.res_C) return 0.f;
    const float* r = ep.res + (long)b * ep.res_fs;
    if (ep.res_mode == RES_DIRECT) return r[((long)oy * Wo + ox) * ep.res_C + c];
    if (ep.res_mode == RES_MAXPOOL) {
        const float* p = r + ((long)(2 * oy) * ep.res_W + 2 * ox) * ep.res_C + c;
        float m0 = fmaxf(p[0], p[ep.res_C]);
        const float* q = p + (long)ep.res_W * ep.res_C;
        float m1 = fmaxf(q[0], q[ep.res_C]);
        return fmaxf(m0, m1);
    }
    // RES_UP2X: TFLite ResizeBilinear, half_pixel_centers, scale 1/2 (see resize2x_kernel)
    float iy = ((float)oy + 0.5f) * 0.5f - 0.5f, ix = ((float)ox + 0.5f) * 0.5f - 0.5f;
    int y0 = max((int)floorf(iy), 0), y1 = min((int)ceilf(iy), ep.res_H - 1);
    int x0 = max((int)floorf(ix), 0), x1 = min((int)ceilf(ix), ep.res_W - 1);
    float dy = iy - (float)y0, dx = ix - (float)x0;
    float p00 = r[((long)y0 * ep.res_W + x0) * ep.res_C + c], p01 = r[((long)y0 * ep.res_W + x1) * ep.res_C + c];
    float p10 = r[((long)y1 * ep.res_W + x0) * ep.res_C + c], p11 = r[((long)y1 * ep.res_W + x1) * ep.res_C + c];
    return p00 * (1 - dy) * (1 - dx) + p10 * dy * (1 - dx) + p01 * (1 - dy) * dx + p11 * dy * dx;
}

// ------------------------------------------------------------------------------------------------ generic conv
// One thread = one output pixel x 4 consecutive output channels.  Threads of a wave that share a pixel read the
// same input scalar (broadcast) and consecutive float4s of the [KH][KW][C][Cop] filter (coalesced, L1/L2 resident).
__global__ __launch_bounds__(256) void conv_generic_kernel(ConvArgs a) {
    const int G = a.Cop >> 2;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)a.B * a.Ho * a.Wo * G;
    if (idx >= total) return;
    int g = (int)(idx % G);
    long p = idx / G;
    int ox = (int)(p % a.Wo);
    long q = p / a.Wo;
    int oy = (int)(q % a.Ho);
    int b = (int)(q / a.Ho);
    const float* in = a.in + (long)b * a.in_fs;
    const bool vec4 = (a.C & 3) == 0 && (a.in_fs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0;  // same accumulation order either way
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < a.KH; ky++) {
        int iy = oy * a.sh - a.pt + ky;
        if (iy < 0 || iy >= a.H) continue;
        for (int kx = 0; kx < a.KW; kx++) {
            int ix = ox * a.sw - a.pl + kx;
            if (ix < 0 || ix >= a.W) continue;
            const float* ip = in + ((long)iy * a.W + ix) * a.C;
            const float4* wp = reinterpret_cast<const float4*>(a.w + ((long)(ky * a.KW + kx) * a.C) * a.Cop) + g;
            int c = 0;
            if (vec4) {  // the pixel's channels as float4s (front / short: 36 -> 42 and 42 -> 48 channels, which no fused kernel takes)
                for (; c + 4 <= a.C; c += 4) {
                    const float4 x = *reinterpret_cast<const float4*>(ip + c);
                    const float4 w0 = wp[(long)c * G], w1 = wp[(long)(c + 1) * G], w2 = wp[(long)(c + 2) * G], w3 = wp[(long)(c + 3) * G];
                    acc.x = fmaf(x.x, w0.x, acc.x); acc.y = fmaf(x.x, w0.y, acc.y); acc.z = fmaf(x.x, w0.z, acc.z); acc.w = fmaf(x.x, w0.w, acc.w);
                    acc.x = fmaf(x.y, w1.x, acc.x); acc.y = fmaf(x.y, w1.y, acc.y); acc.z = fmaf(x.y, w1.z, acc.z); acc.w = fmaf(x.y, w1.w, acc.w);
                    acc.x = fmaf(x.z, w2.x, acc.x); acc.y = fmaf(x.z, w2.y, acc.y); acc.z = fmaf(x.z, w2.z, acc.z); acc.w = fmaf(x.z, w2.w, acc.w);
                    acc.x = fmaf(x.w, w3.x, acc.x); acc.y = fmaf(x.w, w3.y, acc.y); acc.z = fmaf(x.w, w3.z, acc.z); acc.w = fmaf(x.w, w3.w, acc.w);
                }
            }
            for (; c < a.C; c++) {
                float x = ip[c];
                float4 w = wp[(long)c * G];
                acc.x = fmaf(x, w.x, acc.x);
                acc.y = fmaf(x, w.y, acc.y);
                acc.z = fmaf(x, w.z, acc.z);
                acc.w = fmaf(x, w.w, acc.w);
            }
        }
    }
    float v[4] = {acc.x, acc.y, acc.z, acc.w};
    float* op = a.out + (long)b * a.out_fs + ((long)oy * a.Wo + ox) * a.Co;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int c = 4 * g + j;
        if (c < a.Co) {
            float r = v[j] + (a.ep.bias ? a.ep.bias[c] : 0.f);
            const float sk = load_res(a.ep, b, oy, ox, a.Wo, c);
            if (!a.ep.res_after) r += sk;
            r = apply_act(r, a.ep.act, a.ep.alpha, c);
            if (a.ep.res_after) r += sk;
            op[c] = r;
        }
    }
}


// Pointwise convolution to at most 4 output channels (the classifier heads: 48 -> 1 at 48x48 in full_range, 88 -> 2 in front / short):
// one thread = one pixel, its C input channels read as float4s, the [C][4] filter through the scalar cache.  (The generic kernel above
// reads the pixel scalar by scalar: 48 four-byte loads 192 bytes apart per lane.)
__global__ __launch_bounds__(256) void pw_few_kernel(ConvArgs a) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per_frame = (long)a.Ho * a.Wo;
    if (p >= (long)a.B * per_frame) return;
    const int b = (int)(p / per_frame);
    const long q = p - (long)b * per_frame;
    const int oy = (int)(q / a.Wo), ox = (int)(q - (long)oy * a.Wo);
    const float* ip = a.in + (long)b * a.in_fs + q * a.C;
    const float4* __restrict__ w = reinterpret_cast<const float4*>(a.w);  // [C][Cop = 4], uniform
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < a.C; j += 4) {
        const float4 x = *reinterpret_cast<const float4*>(ip + j);
        const float4 w0 = w[j], w1 = w[j + 1], w2 = w[j + 2], w3 = w[j + 3];
        acc.x = fmaf(x.x, w0.x, acc.x); acc.y = fmaf(x.x, w0.y, acc.y); acc.z = fmaf(x.x, w0.z, acc.z); acc.w = fmaf(x.x, w0.w, acc.w);
        acc.x = fmaf(x.y, w1.x, acc.x); acc.y = fmaf(x.y, w1.y, acc.y); acc.z = fmaf(x.y, w1.z, acc.z); acc.w = fmaf(x.y, w1.w, acc.w);
        acc.x = fmaf(x.z, w2.x, acc.x); acc.y = fmaf(x.z, w2.y, acc.y); acc.z = fmaf(x.z, w2.z, acc.z); acc.w = fmaf(x.z, w2.w, acc.w);
        acc.x = fmaf(x.w, w3.x, acc.x); acc.y = fmaf(x.w, w3.y, acc.y); acc.z = fmaf(x.w, w3.z, acc.z); acc.w = fmaf(x.w, w3.w, acc.w);
    }
    const float v[4] = {acc.x, acc.y, acc.z, acc.w};
    float* op = a.out + (long)b * a.out_fs + q * a.Co;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (j < a.Co) {
            float r = v[j] + (a.ep.bias ? a.ep.bias[j] : 0.f);
            const float sk = load_res(a.ep, b, oy, ox, a.Wo, j);
            if (!a.ep.res_after) r += sk;
            r = apply_act(r, a.ep.act, a.ep.alpha, j);
            if (a.ep.res_after) r += sk;
            op[j] = r;
        }
}

// ------------------------------------------------------------------------------------------------ stem convolution
// First layer of every graph: K x K, stride 2, 3 input channels (RGB) -> CO channels, + bias + ReLU/PReLU.
// (back/front 5x5 -> 24, full 3x3 -> 32, landmark 3x3 -> 16, iris 3x3 -> 64; SURVEY.md Appendix A.)
// One thread = one output pixel x all CO channels: the K*K*3 filter taps are wave-uniform, so the weights travel
// through the scalar cache into SGPR operands of v_fmac and every VALU lane does useful work; the input patch comes
// from an LDS tile (zero padded, TF SAME), staged with coalesced row-segment loads.  Output 16 B stores.
// U8: the input is 8UC3 frames (rows of u8_row_bytes bytes); a byte is normalised through a 256-entry table while the tile is filled.
// CS: the output channels are split over CS workgroups per tile (a single-image call, face_detection.rs:205, has 2 - 32 tiles for 256 CUs and
// a thread's K K 3 CO FMAs are one dependent chain of scalar-cache round trips: CO / CS channels per thread shorten it; the per-channel
// arithmetic is the same, so the results are bit-identical).
template <int K, int CO_ALL, bool U8 = false, int CS = 1, int PP = 2>
__global__ __launch_bounds__(256) void stem_conv_kernel(ConvArgs a) {
    constexpr int CO = CO_ALL / CS;
    static_assert(CO * CS == CO_ALL && CO % 4 == 0, "channel split");
    // PP output pixels per thread (rows ly and ly + 8): every scalar-loaded weight pair then feeds PP packed FMAs, which
    // halves the scalar-cache round trips per FMA (they, not the VALU, set the pace at one pixel per thread).  The channel-split form of a
    // single-image call takes PP = 1: twice the workgroups, half the chain.
    constexpr int TW = 32, TH = 8 * PP;                  // output tile
    constexpr int IW = 2 * TW + K - 2, IH = 2 * TH + K - 2;  // input tile (stride 2)
    constexpr int RS = (IW * 3 + 1) & ~1;                // row stride in floats (even: 8-byte aligned float2 reads)
    constexpr int LDSF = (IH * RS > 8 * TW * (CO + 4)) ? IH * RS : 8 * TW * (CO + 4);
    __shared__ __attribute__((aligned(16))) float tile[LDSF];
    const int tid = threadIdx.x;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    int t = blockIdx.x;
    const int cs0 = (t % CS) * CO; t /= CS;   // first output channel of this workgroup
    const int tx0 = (t % tiles_x) * TW; t /= tiles_x;
    const int ty0 = (t % tiles_y) * TH;
    const int b = t / tiles_y;
    const float* in = a.in + (long)b * a.in_fs;
    const int ix0 = tx0 * 2 - a.pl, iy0 = ty0 * 2 - a.pt;
    // All loads of the tile are issued before the first LDS write (a load -> wait -> write loop pays one HBM round trip per
    // iteration).  Thread e owns float e of every tile row (a row is IW * 3 <= 201 consecutive floats of the image): the column
    // arithmetic is done once per thread, a row costs an add and a select — the flat (index -> row, column) mapping this
    // replaces spent a third of the kernel's VALU instructions on divisions and bounds tests.  Loads are unconditional
    // (clamped address, value zeroed afterwards): a predicated load would put a branch around each of them.
    constexpr int ROWF = IW * 3;
    static_assert(ROWF <= 256, "one thread per float of a tile row");
    const int e_col = min(tid, ROWF - 1);
    const int ix = ix0 + e_col / 3;
    const bool x_ok = tid < ROWF && ix >= 0 && ix < a.W;
    float stage[IH];
    if constexpr (U8) {
        __shared__ float lut[256];
        lut[tid] = a.u8_lut[tid];
        const uint8_t* colb = a.in_u8 + (long)b * a.u8_frame_bytes + (long)min(max(ix, 0), a.W - 1) * 3 + (e_col - (e_col / 3) * 3);
        unsigned char raw[IH];
#pragma unroll
        for (int r = 0; r < IH; r++) raw[r] = colb[(long)min(max(iy0 + r, 0), a.H - 1) * a.u8_row_bytes];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < IH; r++) stage[r] = lut[raw[r]];
    } else {
        const float* colp = in + (long)min(max(ix, 0), a.W - 1) * 3 + (e_col - (e_col / 3) * 3);
#pragma unroll
        for (int r = 0; r < IH; r++) stage[r] = colp[(long)min(max(iy0 + r, 0), a.H - 1) * a.W * 3];   // iy0 + r is wave-uniform
    }
    if (tid < ROWF) {
#pragma unroll
        for (int r = 0; r < IH; r++) {
            const int iy = iy0 + r;
            tile[r * RS + tid] = (x_ok && iy >= 0 && iy < a.H) ? stage[r] : 0.f;
        }
    }
    __syncthreads();
    const int lx = tid & (TW - 1), ly = tid / TW;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f acc2[PP][CO / 2];  // two output channels per v_pk_fma_f32 (the weight pair is an SGPR-pair operand)
#pragma unroll
    for (int p = 0; p < PP; p++)
#pragma unroll
        for (int o = 0; o < CO / 2; o++) acc2[p][o] = v2f{0.f, 0.f};
    const float* __restrict__ w = a.w + cs0;  // [K][K][3][Cop], uniform -> scalar loads
    const int Cop = (CO_ALL + 3) & ~3;
    // Not unrolled beyond 3 taps: the weights of a tap are 24..64 SGPRs; letting the compiler hoist all K*K*3 taps'
    // scalar loads would spill SGPRs into VGPR lanes.  Occupancy (<= 64 VGPRs) hides the scalar-load latency instead.
#pragma unroll 1
    for (int ky = 0; ky < K; ky++) {
        const float* row = tile + (2 * ly + ky) * RS + 2 * lx * 3;
        const float* wk = w + (long)(ky * K * 3) * Cop;
#pragma unroll 3
        for (int e = 0; e < K * 3; e++) {
            float xv[PP];
#pragma unroll
            for (int p = 0; p < PP; p++) xv[p] = row[p * 16 * RS + e];
            const float* we = wk + (long)e * Cop;
#pragma unroll
            for (int o = 0; o < CO / 2; o++)
#pragma unroll
                for (int p = 0; p < PP; p++) acc2[p][o] = __builtin_elementwise_fma(v2f{xv[p], xv[p]}, v2f{we[2 * o], we[2 * o + 1]}, acc2[p][o]);
        }
    }
    // Epilogue through LDS: a thread owns one pixel (CO floats), but 16-byte stores at a CO*4-byte lane stride reach
    // HBM as partial 32-byte sectors (measured 2.85x WRITE_SIZE).  Re-tile so that each wave-instruction writes
    // 1 KiB of consecutive addresses: the tile's rows are contiguous runs of TW*CO floats in the NHWC output.
    float* otile = tile;  // [8*TW][CO + 4] (pad keeps the float4 writes of consecutive pixels on distinct banks)
    constexpr int OS = CO + 4;
    // branch-free activation: act(v) = min(max(v,0) + slope * min(v,0), hi) with slope 0 (ReLU / ReLU6), alpha (PReLU) or 1
    // (none).  A switch per element costs a scalar load + wait + branches for each of the CO channels of every wave,
    // which took longer than the 900 packed FMAs of the 5x5 stem.
    const bool prelu = a.ep.act == ACT_PRELU;
    const float base_slope = a.ep.act == ACT_NONE ? 1.f : 0.f, hi = a.ep.act == ACT_RELU6 ? 6.f : INFINITY;
    const float* __restrict__ bias = a.ep.bias + cs0;
    const float* __restrict__ al = (prelu ? a.ep.alpha : a.ep.bias) + cs0;  // always a readable array: no branch around the loads
    constexpr int C4 = CO / 4;
    const bool relu = a.ep.act == ACT_RELU;  // wave-uniform: the three detector stems; one packed add + one packed max per channel pair
#pragma unroll
    for (int p = 0; p < PP; p++) {
        __syncthreads();  // all reads of the input tile (p = 0) / of the previous half's output image are done; reuse the LDS
        if (relu) {
#pragma unroll
            for (int o = 0; o < CO; o += 4) {
                const v2f r0 = __builtin_elementwise_max(acc2[p][o >> 1] + v2f{bias[o], bias[o + 1]}, v2f{0.f, 0.f});
                const v2f r1 = __builtin_elementwise_max(acc2[p][(o >> 1) + 1] + v2f{bias[o + 2], bias[o + 3]}, v2f{0.f, 0.f});
                *reinterpret_cast<float4*>(otile + tid * OS + o) = make_float4(r0.x, r0.y, r1.x, r1.y);
            }
        } else {
#pragma unroll
            for (int o = 0; o < CO; o += 4) {
                float r[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float acc = (e & 1) ? acc2[p][(o + e) >> 1].y : acc2[p][(o + e) >> 1].x;
                    const float v = acc + bias[o + e];
                    const float sl = prelu ? al[o + e] : base_slope;
                    r[e] = fminf(fmaxf(v, 0.f) + sl * fminf(v, 0.f), hi);
                }
                *reinterpret_cast<float4*>(otile + tid * OS + o) = make_float4(r[0], r[1], r[2], r[3]);
            }
        }
        __syncthreads();
        for (int i = tid; i < 8 * TW * C4; i += 256) {
            int px = i / C4, c4 = i - px * C4;
            int y = ty0 + 8 * p + px / TW, x = tx0 + (px & (TW - 1));
            if (y < a.Ho && x < a.Wo)
                *reinterpret_cast<float4*>(a.out + (long)b * a.out_fs + ((long)y * a.Wo + x) * CO_ALL + cs0 + 4 * c4) =
                    *reinterpret_cast<const float4*>(otile + px * OS + 4 * c4);
        }
    }
}

// ---- the 5 x 5 stem (BackCamera 256x256 -> 128x128x24, Short / Front 128x128 -> 64x64x24) on the matrix cores, round 6.
// stem_conv_kernel does its 1 800 FMAs per output pixel as packed VALU FMAs with the weights as scalar operands: 83 TFLOP/s at 256 frames, of the 104 - 116 the
// packed FMAs reach on this part (0.18 ms of BASELINE config 2's 1.30).  v_mfma_f32_4x4x1_16b_f32 is 16 independent (4 x 1)(1 x 4) outer products with
// D[i][lane] = A[block, i] * B[lane] + C[i][lane] (strip_kernels.hip uses it the same way): with lane = OUTPUT PIXEL, B = one value of the pixel's 5 x 5 x 3
// window and A = four output-channel weights W[4 t .. 4 t + 3][k], broadcast from one block of a weight register (cbsz = 4, abid = block), D = four output
// channels of 64 pixels.  The whole filter — 75 x 24 weights — is 29 registers of 64 lanes, loaded once per wave; a tile of 64 pixels is 75 x 6 = 450 MFMAs
// and nothing else but its loads: five rows of 15 consecutive floats per lane (the window's row: 3 x dwordx4 + dwordx3 buffer loads, the next row's
// in flight while this row's 90 MFMAs issue).  Rows above / below the picture are whole loads outside the buffer resource (zeros); the pixel left of
// column 0 and the two right of column W - 1 are handled in the two edge lanes, which load a window shifted into the picture and move the values
// over (every load is wholly inside or wholly outside the resource).  Each output is the same k-sequential f32 FMA chain from 0, bias added behind it,
// as in stem_conv_kernel: the results are bit-identical (tests/test_gpu_parity.py::test_mfma_stem_bit_equal_to_the_valu_stem).
// Measured (BackCamera, 256 frames; profiles/r06_stem_mfma.txt): 0.182 -> 0.165 ms (one frame: 8.9 -> 7.0 us).  What bounds it now is memory, not the matrix cores: without its MFMAs the
// kernel takes 0.140 ms (201 MB in + 403 MB out = 4.3 TB/s), without its stores 0.147, its 29.5 M MFMAs alone are 0.105 ms at the 8.2 cycles
// tools/probes/mfma4_probe.hip measures for this stream; the tensor it writes is read once more by the first row pipeline.
typedef float sv4f __attribute__((ext_vector_type(4)));
typedef unsigned su4 __attribute__((ext_vector_type(4)));
typedef unsigned su3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ sv4f stem_mfma4(float a, float b, sv4f c, int blk) {   // (the builtin wants the block as a literal: the switch folds away once the loops are unrolled)
    switch (blk) {
#define MI_SM4(N) case N: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, N, 0);
        MI_SM4(0) MI_SM4(1) MI_SM4(2) MI_SM4(3) MI_SM4(4) MI_SM4(5) MI_SM4(6) MI_SM4(7) MI_SM4(8) MI_SM4(9) MI_SM4(10) MI_SM4(11) MI_SM4(12) MI_SM4(13) MI_SM4(14)
#undef MI_SM4
        default: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 15, 0);
    }
}

template <int N, class F, int... KS>
__device__ __forceinline__ void dfor_tiles_impl(F&& f, std::integer_sequence<int, KS...>) { (f(std::integral_constant<int, KS>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void dfor_tiles(F&& f) { dfor_tiles_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <bool RELU>
__global__ __launch_bounds__(256) void stem_mfma_kernel(ConvArgs a, int tiles, int tpr) {
    constexpr int KW3 = 15, NK = 75, CQ = 6, NA = (NK * CQ + 15) / 16;   // 29 weight registers
    const int lane = threadIdx.x & 63;
    // Workgroup i runs on XCD i % 8, each with an L2 of its own: the workgroups of one XCD take CONSECUTIVE tiles (the rows of one stretch of a frame: a picture
    // row is read by two or three output rows), not every eighth group of four — interleaved, an XCD fetched 7 picture rows per 2 output rows
    const int G = (int)gridDim.x, wg = (G & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
    const int wv = __builtin_amdgcn_readfirstlane(wg * 4 + (int)(threadIdx.x >> 6)), nwv = G * 4;
    // weight register r, lanes 4 (n % 16) + i = W[4 t + i][k] for n = 16 r + ... = k CQ + t  (a.w: [k][24])
    float wa[NA];
#pragma unroll
    for (int r = 0; r < NA; r++) {
        const int n = 16 * r + (lane >> 2), k = n / CQ, t = n - CQ * k;
        wa[r] = n < NK * CQ ? a.w[k * 24 + 4 * t + (lane & 3)] : 0.f;
    }
    const float hi = a.ep.act == ACT_RELU6 ? 6.f : INFINITY, base_slope = a.ep.act == ACT_NONE ? 1.f : 0.f;
    const bool prelu = a.ep.act == ACT_PRELU;
    // bias and slopes: wave-uniform, read ONCE into scalar registers (read behind the row loop's stores they become vector loads the compiler orders behind
    // each store: six store round trips per tile)
    float bs[24], sl[RELU ? 1 : 24];
    {
        const float* __restrict__ bias = a.ep.bias;
        const float* __restrict__ al = prelu ? a.ep.alpha : a.ep.bias;   // always a readable array
#pragma unroll
        for (int i = 0; i < 24; i++) bs[i] = bias[i];
        if constexpr (!RELU) {
#pragma unroll
            for (int i = 0; i < 24; i++) sl[i] = prelu ? al[i] : base_slope;
        }
    }
    const int rowb = a.W * 12;                                       // bytes of a picture row
    __shared__ __attribute__((aligned(16))) float otile[4 * 64 * 24];
    // A wave's row loads form one stream over its tiles — five rows per tile, the next row's loads issued before this row's 90 MFMAs, the NEXT TILE's first row
    // before this tile's last 90: without that every tile began with an exposed memory round trip that the other waves of the SIMD, in step with this
    // one, did not cover (0.220 ms with, 0.147 ms without the MFMAs, 0.105 ms of MFMAs).  NB buffers in turn, a row's loads issued NB - 1 rows ahead; five
    // rows per tile, so the buffers' roles rotate from tile to tile (P).
    struct Tile { int b, y, half, px, lane_off; bool left, right; };
    auto setup = [&](int t) {
        Tile c;
        c.b = t / (a.Ho * tpr);
        const int rem = t - c.b * (a.Ho * tpr);
        c.y = rem / tpr; c.half = rem - c.y * tpr;                   // wave-uniform
        c.px = 64 * c.half + lane;
        c.left = c.px == 0; c.right = c.px == a.Wo - 1;
        // the window of pixel px starts at column 2 px - 1: the edge lanes read columns 0 .. 4 / W - 5 .. W - 1 instead
        c.lane_off = (2 * c.px - 1 + (c.left ? 1 : 0) - (c.right ? 2 : 0)) * 12;
        return c;
    };
    auto load_row = [&](const Tile& c, int ky, float (&x)[KW3]) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in + (long)c.b * a.in_fs), 0, a.H * rowb, 0x00020000);
        const int vo = (2 * c.y - 1 + ky) * rowb + c.lane_off;   // row -1: negative = far outside; rows >= H: outside
        const su4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0), q1 = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 16, 0, 0),
                  q2 = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 32, 0, 0);
        const su3 q3 = __builtin_amdgcn_raw_buffer_load_b96(rs, vo + 48, 0, 0);
        x[0] = __uint_as_float(q0.x); x[1] = __uint_as_float(q0.y); x[2] = __uint_as_float(q0.z); x[3] = __uint_as_float(q0.w);
        x[4] = __uint_as_float(q1.x); x[5] = __uint_as_float(q1.y); x[6] = __uint_as_float(q1.z); x[7] = __uint_as_float(q1.w);
        x[8] = __uint_as_float(q2.x); x[9] = __uint_as_float(q2.y); x[10] = __uint_as_float(q2.z); x[11] = __uint_as_float(q2.w);
        x[12] = __uint_as_float(q3.x); x[13] = __uint_as_float(q3.y); x[14] = __uint_as_float(q3.z);
    };
    constexpr int NB = 2;   // (three buffers, loads two rows ahead: the same 0.165 ms)
    float xb[NB][KW3];
    auto tile = [&](auto pc, const Tile& c, const Tile& nx) {
        constexpr int P = decltype(pc)::value;
        sv4f D[CQ];
#pragma unroll
        for (int q = 0; q < CQ; q++) D[q] = sv4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 5; ky++) {
            if (ky + NB - 1 < 5) load_row(c, ky + NB - 1, xb[(ky + NB - 1 + P) % NB]);
            else load_row(nx, ky + NB - 1 - 5, xb[(ky + NB - 1 + P) % NB]);   // (the last tile loads its own first rows again: no branch around a load)
            float (&x)[KW3] = xb[(ky + P) % NB];
            if (c.half == 0) {          // wave-uniform: lane 0 holds columns 0 .. 4, its window is (zero), 0 .. 3
#pragma unroll
                for (int e = KW3 - 1; e >= 0; e--) x[e] = c.left ? (e >= 3 ? x[e - 3] : 0.f) : x[e];
            }
            if (c.half == tpr - 1) {    // lane 63 holds columns W - 5 .. W - 1, its window is W - 3 .. W - 1, (zero), (zero)
#pragma unroll
                for (int e = 0; e < KW3; e++) x[e] = c.right ? (e < 9 ? x[e + 6] : 0.f) : x[e];
            }
#pragma unroll
            for (int e = 0; e < KW3; e++)
#pragma unroll
                for (int q = 0; q < CQ; q++) {
                    const int n = (ky * KW3 + e) * CQ + q;
                    D[q] = stem_mfma4(wa[n >> 4], x[e], D[q], n & 15);
                }
        }
        // The tile's 64 x 24 results are 6 KB of consecutive addresses, but a lane holds ONE pixel's 96 bytes: stored from the registers, each of the six
        // 16-byte store instructions touches every line of the tile (partial sectors at the memory side: loads + stores alone took 0.22 ms).  Through LDS
        // instead: every store instruction writes 1 KB of consecutive addresses.
        float* ot = otile + (threadIdx.x >> 6) * (64 * 24);
#pragma unroll
        for (int q = 0; q < CQ; q++) {
            float r[4] = {D[q].x, D[q].y, D[q].z, D[q].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float v = r[i] + bs[4 * q + i];
                if constexpr (RELU) r[i] = fmaxf(v, 0.f);
                else r[i] = fminf(fmaxf(v, 0.f) + sl[4 * q + i] * fminf(v, 0.f), hi);
            }
            *reinterpret_cast<float4*>(ot + lane * 24 + 4 * q) = make_float4(r[0], r[1], r[2], r[3]);
        }
        __builtin_amdgcn_wave_barrier();   // (the tile is this wave's own: LDS operations of a wave execute in order)
        float* op = a.out + (long)c.b * a.out_fs + ((long)c.y * a.Wo + 64 * c.half) * 24;
#pragma unroll
        for (int q = 0; q < CQ; q++)
            *reinterpret_cast<float4*>(op + (q * 64 + lane) * 4) = *reinterpret_cast<const float4*>(ot + (q * 64 + lane) * 4);
        __builtin_amdgcn_wave_barrier();
    };
    if (wv >= tiles) return;
    Tile tl[2];
    tl[0] = setup(wv);
#pragma unroll
    for (int r = 0; r < NB - 1; r++) load_row(tl[0], r, xb[r]);
    // tile k of this wave starts at row 5 k of the stream: P = 5 k mod NB
    for (int t = wv;;) {
        bool done = false;
        dfor_tiles<NB>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (done) return;
            const int tn = t + nwv;
            tl[(k + 1) & 1] = setup(min(tn, tiles - 1));
            tile(std::integral_constant<int, (5 * k) % NB>{}, tl[k & 1], tl[(k + 1) & 1]);
            if (tn >= tiles) done = true;
            t = tn;
        });
        if (done) break;
        if constexpr (NB & 1) { const Tile sw = tl[0]; tl[0] = tl[1]; tl[1] = sw; }   // (an odd number of tiles per round: the roles of tl[] swap)
    }
}

static bool stem_mfma_applicable(const ConvArgs& a) {
    static const bool off = getenv("MI_NO_STEM_MFMA") != nullptr;   // tuning aid
    if (off || a.no_mfma || a.in_u8 || a.KH != 5 || a.KW != 5 || a.C != 3 || a.Co != 24 || a.Cop != 24 || a.sh != 2 || a.sw != 2 || a.pt != 1 || a.pl != 1) return false;
    if (a.W != 2 * a.Wo || a.H != 2 * a.Ho || a.Wo % 64 || a.W < 8 || !a.ep.bias || a.ep.res_mode != RES_NONE) return false;
    if ((reinterpret_cast<uintptr_t>(a.in) & 15) || (a.in_fs & 3) || (reinterpret_cast<uintptr_t>(a.out) & 15) || (a.out_fs & 3)) return false;
    if ((long)a.H * a.W * 12 > 0x7fffffffL) return false;
    return (long)a.B * a.Ho * (a.Wo / 64) >= 64;   // tiles of 64 pixels.  (One BackCamera frame = 256 tiles on 64 workgroups: 7.0 us against the split packed-FMA form's 8.9, 8.4 against 15.7 at four frames)
}

static int launch_stem_mfma(const ConvArgs& a, hipStream_t s) {
    const int tpr = a.Wo / 64, tiles = a.B * a.Ho * tpr;
    const int per_cu = 4;   // workgroups of four waves (2 .. 6 per CU measured alike: 0.165 - 0.168 ms)
    unsigned grid = (unsigned)std::min<long>((tiles + 3) / 4, (long)per_cu * device_cu_count());
    if (grid >= 64) grid &= ~7u;   // (a multiple of the XCD count: the kernel's tile order assumes it)
    if (a.ep.act == ACT_RELU) return (int)launch_kernel(stem_mfma_kernel<true>, dim3(grid), dim3(256), 0, s, a, tiles, tpr);
    return (int)launch_kernel(stem_mfma_kernel<false>, dim3(grid), dim3(256), 0, s, a, tiles, tpr);
}

template <int K, int CO>
static int launch_stem(const ConvArgs& a, hipStream_t s) {
    constexpr int TH = 16;  // rows of a tile (two output pixels per thread)
    unsigned tiles = (unsigned)(((a.Wo + 31) / 32) * ((a.Ho + TH - 1) / TH));
    if (a.in_u8) return (int)launch_kernel(stem_conv_kernel<K, CO, true>, dim3(tiles * (unsigned)a.B), dim3(256), 0, s, a);
    // a handful of frames: 8 output channels per workgroup (16 of the iris network's 64), CS times the workgroups
    constexpr int CS = CO == 64 ? 4 : CO / 8;
    const unsigned tiles1 = (unsigned)(((a.Wo + 31) / 32) * ((a.Ho + 7) / 8));   // 8-row tiles: one output pixel per thread
    if (tiles1 * (unsigned)a.B * CS <= 256u) return (int)launch_kernel(stem_conv_kernel<K, CO, false, CS, 1>, dim3(tiles1 * (unsigned)a.B * CS), dim3(256), 0, s, a);
    if (tiles * (unsigned)a.B * CS <= 256u) return (int)launch_kernel(stem_conv_kernel<K, CO, false, CS>, dim3(tiles * (unsigned)a.B * CS), dim3(256), 0, s, a);
    return (int)launch_kernel(stem_conv_kernel<K, CO, false>, dim3(tiles * (unsigned)a.B), dim3(256), 0, s, a);
}

// true when (and how) the specialised stem kernel takes this convolution
static bool stem_applicable(const ConvArgs& a) {
    if (a.C != 3 || a.sh != 2 || a.sw != 2 || a.KH != a.KW || !a.ep.bias || a.ep.res_mode != RES_NONE) return false;
    if ((reinterpret_cast<uintptr_t>(a.out) & 15) || (a.out_fs & 3)) return false;
    return (a.KH == 5 && a.Co == 24) || (a.KH == 3 && (a.Co == 16 || a.Co == 32 || a.Co == 64));
}

static bool pw_few_applicable(const ConvArgs& a) {
    return a.KH == 1 && a.KW == 1 && a.sh == 1 && a.sw == 1 && a.Cop == 4 && a.C % 4 == 0 && a.H == a.Ho && a.W == a.Wo && (a.in_fs & 3) == 0 &&
           (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0;
}
bool conv_takes_u8(const ConvArgs& a) { return stem_applicable(a); }
const char* conv_kernel_label(const ConvArgs& a) { return stem_applicable(a) ? (stem_mfma_applicable(a) ? "stem_mfma_kernel" : "stem_conv_kernel") : (pw_few_applicable(a) ? "pw_few_kernel" : "conv_generic_kernel"); }

int launch_conv(const ConvArgs& a, void* stream) {
    if (stem_applicable(a)) {
        hipStream_t st = (hipStream_t)stream;
        if (stem_mfma_applicable(a)) return launch_stem_mfma(a, st);
        if (a.KH == 5) return launch_stem<5, 24>(a, st);
        if (a.Co == 16) return launch_stem<3, 16>(a, st);
        if (a.Co == 32) return launch_stem<3, 32>(a, st);
        return launch_stem<3, 64>(a, st);
    }
    long total = (long)a.B * a.Ho * a.Wo * (a.Cop >> 2);
    if (total <= 0) return 0;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (pw_few_applicable(a)) return (int)launch_kernel(pw_few_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return (int)launch_kernel(conv_generic_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
}

// ------------------------------------------------------------------------------------------------ depthwise 3x3
// One thread = one output pixel x V consecutive channels (V = 4/2/1 by alignment).  HBM-bound: 9 MAC per 8 bytes.
template <int V>
__global__ __launch_bounds__(256) void dw_kernel(DwArgs a) {
    const int G = a.C / V;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)a.B * a.Ho * a.Wo * G;
    if (idx >= total) return;
    int g = (int)(idx % G);
    long p = idx / G;
    int ox = (int)(p % a.Wo);
    long q = p / a.Wo;
    int oy = (int)(q % a.Ho);
    int b = (int)(q / a.Ho);
    const float* in = a.in + (long)b * a.in_fs + g * V;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; j++) acc[j] = 0.f;
    for (int ky = 0; ky < a.KH; ky++) {
        int iy = oy * a.sh - a.pt + ky;
        if (iy < 0 || iy >= a.H) continue;
        for (int kx = 0; kx < a.KW; kx++) {
            int ix = ox * a.sw - a.pl + kx;
            if (ix < 0 || ix >= a.W) continue;
            const float* ip = in + ((long)iy * a.W + ix) * a.C;
            const float* wp = a.w + (long)(ky * a.KW + kx) * a.C + g * V;
#pragma unroll
            for (int j = 0; j < V; j++) acc[j] = fmaf(ip[j], wp[j], acc[j]);
        }
    }
    float* op = a.out + (long)b * a.out_fs + ((long)oy * a.Wo + ox) * a.C + g * V;
#pragma unroll
    for (int j = 0; j < V; j++) {
        int c = g * V + j;
        float r = acc[j] + (a.ep.bias ? a.ep.bias[c] : 0.f);
        r += load_res(a.ep, b, oy, ox, a.Wo, c);
        op[j] = apply_act(r, a.ep.act, a.ep.alpha, c);
    }
}

int launch_dw(const DwArgs& a, void* stream) {
    int V = (a.C % 4 == 0) ? 4 : (a.C % 2 == 0 ? 2 : 1);
    long total = (long)a.B * a.Ho * a.Wo * (a.C / V);
    if (total <= 0) return 0;
    unsigned blocks = (unsigned)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (V == 4) return (int)launch_kernel(dw_kernel<4>, dim3(blocks), dim3(256), 0, s, a);
    if (V == 2) return (int)launch_kernel(dw_kernel<2>, dim3(blocks), dim3(256), 0, s, a);
    return (int)launch_kernel(dw_kernel<1>, dim3(blocks), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------ frame-batched head GEMM
// out[b][n] = act(bias[n] + sum_k x[b][k] * W[n][k]) on v_mfma_f32_32x32x2_f32: one wave per 32 frames x 32 outputs.  The
// contraction index is split in two halves (lanes 0-31 take k in [0, K/2), lanes 32-63 take [K/2, K)), so every lane reads
// its operands as float4 of four consecutive k: four MFMAs per pair of 16-byte loads, both operands straight from L2.
typedef float hg_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__global__ __launch_bounds__(256) void head_gemm_kernel(HeadGemmArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * 32, b0 = blockIdx.y * 32;
    if (n0 >= a.N) return;  // whole wave
    const int r = lane & 31, h = lane >> 5, K2 = a.K >> 1;
    const float* xp = a.in + (long)min(b0 + r, a.B - 1) * a.in_fs + h * K2;   // rows past the batch / the outputs re-read the
    const float* wp = a.w + (long)min(n0 + r, a.N - 1) * a.K + h * K2;        // last valid one; their results are not stored
    hg_f32x16 D;
#pragma unroll
    for (int e = 0; e < 16; e++) D[e] = 0.f;
    float4 xa = ld4(xp), wa = ld4(wp);
    for (int j = 0; j < K2; j += 4) {
        const int jn = min(j + 4, K2 - 4);
        const float4 xn = ld4(xp + jn), wn = ld4(wp + jn);  // next chunk in flight under this chunk's MFMAs
        D = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, wa.x, D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.y, wa.y, D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.z, wa.z, D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.w, wa.w, D, 0, 0, 0);
        xa = xn; wa = wn;
    }
    // D[v]: frame b0 + 4 * h + 8 * (v / 4) + v % 4, output n0 + r
    const int n = n0 + r;
    if (n >= a.N) return;
    const float bias = a.bias ? a.bias[n] : 0.f;
    const float slope = a.act == ACT_PRELU ? a.alpha[n] : (a.act == ACT_NONE ? 1.f : 0.f);
    const float hi = a.act == ACT_RELU6 ? 6.f : INFINITY;
#pragma unroll
    for (int v = 0; v < 16; v++) {
        const int b = b0 + 4 * h + 8 * (v >> 2) + (v & 3);
        if (b >= a.B) continue;
        const float t = D[v] + bias;
        a.out[(long)b * a.out_fs + n] = fminf(fmaxf(t, 0.f) + slope * fminf(t, 0.f), hi);
    }
}

// The same product for the one to four frames of a single-image call (face_landmark.rs:265, iris_landmark.rs:203): a 32-frame MFMA tile with one live
// frame is 256 dependent 64-cycle MFMAs for the iris network's K = 512 (17 us per head, two heads in line behind the band launch).  Here a wave owns
// one (output, frame): its lanes read 16 bytes of x and of W[n] each per round (coalesced), four partial sums per lane, then a butterfly over the
// wave — a fixed order, so run-to-run bit-identical (it is not the MFMA's order: within the raw tolerance of the batched form, like every small-batch form).
__global__ __launch_bounds__(256) void head_dot_kernel(HeadGemmArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave, b = blockIdx.y;
    if (n >= a.N) return;  // whole wave
    const float4* x = reinterpret_cast<const float4*>(a.in + (long)b * a.in_fs);
    const float4* w = reinterpret_cast<const float4*>(a.w + (long)n * a.K);
    const int K4 = a.K >> 2;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int j = lane; j < K4; j += 64) {
        const float4 xv = x[j], wv = w[j];
        s0 = fmaf(xv.x, wv.x, s0); s1 = fmaf(xv.y, wv.y, s1); s2 = fmaf(xv.z, wv.z, s2); s3 = fmaf(xv.w, wv.w, s3);
    }
    float sum = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) {
        const float t = sum + (a.bias ? a.bias[n] : 0.f);
        const float slope = a.act == ACT_PRELU ? a.alpha[n] : (a.act == ACT_NONE ? 1.f : 0.f);
        const float hi = a.act == ACT_RELU6 ? 6.f : INFINITY;
        a.out[(long)b * a.out_fs + n] = fminf(fmaxf(t, 0.f) + slope * fminf(t, 0.f), hi);
    }
}

bool head_gemm_supports(int K, int N) { return K >= 16 && (K % 8) == 0 && N >= 1; }

int launch_head_gemm(const HeadGemmArgs& a, void* stream) {
    if (!head_gemm_supports(a.K, a.N) || a.B < 1 || (a.in_fs & 3) || (reinterpret_cast<uintptr_t>(a.in) & 15) || (reinterpret_cast<uintptr_t>(a.w) & 15))
        return (int)hipErrorInvalidValue;
    if (a.B <= 4) return (int)launch_kernel(head_dot_kernel, dim3((unsigned)((a.N + 3) / 4), (unsigned)a.B), dim3(256), 0, (hipStream_t)stream, a);
    const unsigned tiles_n = (unsigned)((a.N + 31) / 32);
    return (int)launch_kernel(head_gemm_kernel, dim3((tiles_n + 3) / 4, (unsigned)((a.B + 31) / 32)), dim3(256), 0, (hipStream_t)stream, a);
}

// ------------------------------------------------------------------------------------------------ element-wise fallbacks
__global__ void add_kernel(EltArgs a) {  // out = act(a + b), same shape
    long per = (long)a.H * a.W * a.C;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per * a.B) return;
    int b = (int)(idx / per);
    long e = idx % per;
    float v = a.a[(long)b * a.a_fs + e] + a.b[(long)b * a.b_fs + e];
    a.out[(long)b * a.out_fs + e] = apply_act(v, a.act, a.alpha, (int)(e % a.C));
}
__global__ void act_kernel(EltArgs a) {
    long per = (long)a.H * a.W * a.C;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per * a.B) return;
    int b = (int)(idx / per);
    long e = idx % per;
    a.out[(long)b * a.out_fs + e] = apply_act(a.a[(long)b * a.a_fs + e], a.act, a.alpha, (int)(e % a.C));
}
// MAX_POOL_2D: p0 = filter_h, p1 = filter_w, p2 = stride_h, p3 = stride_w; SAME/VALID pads folded into Ho/Wo + H/W clip
__global__ void maxpool_kernel(EltArgs a, int pt, int pl) {
    long per = (long)a.Ho * a.Wo * a.C;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per * a.B) return;
    int b = (int)(idx / per);
    long e = idx % per;
    int c = (int)(e % a.C);
    int ox = (int)((e / a.C) % a.Wo), oy = (int)(e / ((long)a.C * a.Wo));
    float m = -INFINITY;
    for (int ky = 0; ky < a.p0; ky++) {
        int iy = oy * a.p2 - pt + ky;
        if (iy < 0 || iy >= a.H) continue;
        for (int kx = 0; kx < a.p1; kx++) {
            int ix = ox * a.p3 - pl + kx;
            if (ix < 0 || ix >= a.W) continue;
            m = fmaxf(m, a.a[(long)b * a.a_fs + ((long)iy * a.W + ix) * a.C + c]);
        }
    }
    a.out[(long)b * a.out_fs + e] = m;
}
// PAD with zeros: p0 = top, p1 = left, p2 = channel-before; output dims Ho, Wo, Co
__global__ void pad_kernel(EltArgs a) {
    long per = (long)a.Ho * a.Wo * a.Co;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per * a.B) return;
    int b = (int)(idx / per);
    long e = idx % per;
    int c = (int)(e % a.Co) - a.p2;
    int x = (int)((e / a.Co) % a.Wo) - a.p1, y = (int)(e / ((long)a.Co * a.Wo)) - a.p0;
    float v = 0.f;
    if (c >= 0 && c < a.C && x >= 0 && x < a.W && y >= 0 && y < a.H) v = a.a[(long)b * a.a_fs + ((long)y * a.W + x) * a.C + c];
    a.out[(long)b * a.out_fs + e] = v;
}
// RESIZE_BILINEAR (half_pixel_centers = p0, align_corners = p1), arbitrary sizes; TFLite reference formula.
__global__ void resize_kernel(EltArgs a) {
    long per = (long)a.Ho * a.Wo * a.C;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per * a.B) return;
    int b = (int)(idx / per);
    long e = idx % per;
    int c = (int)(e % a.C);
    int ox = (int)((e / a.C) % a.Wo), oy = (int)(e / ((long)a.C * a.Wo));
    float hs = (a.p1 && a.Ho > 1) ? (float)(a.H - 1) / (float)(a.Ho - 1) : (float)a.H / (float)a.Ho;
    float ws = (a.p1 && a.Wo > 1) ? (float)(a.W - 1) / (float)(a.Wo - 1) : (float)a.W / (float)a.Wo;
    float iy = a.p0 ? ((float)oy + 0.5f) * hs - 0.5f : (float)oy * hs;
    float ix = a.p0 ? ((float)ox + 0.5f) * ws - 0.5f : (float)ox * ws;
    int y0 = max((int)floorf(iy), 0), y1 = min((int)ceilf(iy), a.H - 1);
    int x0 = max((int)floorf(ix), 0), x1 = min((int)ceilf(ix), a.W - 1);
    float dy = iy - (float)y0, dx = ix - (float)x0;
    const float* r = a.a + (long)b * a.a_fs;
    float p00 = r[((long)y0 * a.W + x0) * a.C + c], p01 = r[((long)y0 * a.W + x1) * a.C + c];
    float p10 = r[((long)y1 * a.W + x0) * a.C + c], p11 = r[((long)y1 * a.W + x1) * a.C + c];
    a.out[(long)b * a.out_fs + e] = p00 * (1 - dy) * (1 - dx) + p10 * dy * (1 - dx) + p01 * (1 - dy) * dx + p11 * dy * dx;
}
// DEPTH_TO_SPACE: p0 = block size
__global__ void d2s_kernel(EltArgs a) {
    long per = (long)a.Ho * a.Wo * a.Co;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per * a.B) return;
    int b = (int)(idx / per);
    long e = idx % per;
    int c = (int)(e % a.Co);
    int ox = (int)((e / a.Co) % a.Wo), oy = (int)(e / ((long)a.Co * a.Wo));
    int bs = a.p0;
    int iy = oy / bs, ix = ox / bs, ic = ((oy % bs) * bs + (ox % bs)) * a.Co + c;
    a.out[(long)b * a.out_fs + e] = a.a[(long)b * a.a_fs + ((long)iy * a.W + ix) * a.C + ic];
}
// strided frame copy: per-frame `C` floats (H = W = 1)
__global__ void copy_kernel(EltArgs a) {
    long per = (long)a.C;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per * a.B) return;
    int b = (int)(idx / per);
    long e = idx % per;
    a.out[(long)b * a.out_fs + e] = a.a[(long)b * a.a_fs + e];
}

static inline unsigned nblocks(long n) { return (unsigned)((n + 255) / 256); }
int launch_add(const EltArgs& a, void* s) {
    long n = (long)a.B * a.H * a.W * a.C;
    return n > 0 ? (int)launch_kernel(add_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)s, a) : 0;
}
int launch_act(const EltArgs& a, void* s) {
    long n = (long)a.B * a.H * a.W * a.C;
    return n > 0 ? (int)launch_kernel(act_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)s, a) : 0;
}
int launch_maxpool(const EltArgs& a, void* s) {
    long n = (long)a.B * a.Ho * a.Wo * a.C;
    // TF SAME pads (only non-zero for odd inputs): total = max(0,(out-1)*stride + filter - in), before = total/2
    int tph = (a.Ho - 1) * a.p2 + a.p0 - a.H, tpw = (a.Wo - 1) * a.p3 + a.p1 - a.W;
    int pt = tph > 0 ? tph / 2 : 0, pl = tpw > 0 ? tpw / 2 : 0;
    return n > 0 ? (int)launch_kernel(maxpool_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)s, a, pt, pl) : 0;
}
int launch_padc(const EltArgs& a, void* s) {
    long n = (long)a.B * a.Ho * a.Wo * a.Co;
    return n > 0 ? (int)launch_kernel(pad_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)s, a) : 0;
}
int launch_resize2x(const EltArgs& a, void* s) {
    long n = (long)a.B * a.Ho * a.Wo * a.C;
    return n > 0 ? (int)launch_kernel(resize_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)s, a) : 0;
}
int launch_depth_to_space(const EltArgs& a, void* s) {
    long n = (long)a.B * a.Ho * a.Wo * a.Co;
    return n > 0 ? (int)launch_kernel(d2s_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)s, a) : 0;
}
int launch_copy_strided(const EltArgs& a, void* s) {
    long n = (long)a.B * a.C;
    return n > 0 ? (int)launch_kernel(copy_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)s, a) : 0;
}

// ------------------------------------------------------------------------------------------------ SSD post-processing
// One 256-thread workgroup per frame.  Restates, on the device, the tail of FaceDetection::infer
// (face_detection.rs:259-265): decode_boxes (269-296), get_sigmoid_score (300-314), convert_to_detections
// (317-362), non_maximum_suppression(weighted) (nms.rs:56-144) and detection_letterbox_removal (transform.rs:115-142).
// Ordering/dtype rules kept bit-for-bit: true f32 division by `scale`, score clamp to
// +-80, `sigmoid(x) > 0.5` on the f32 sigmoid, stable descending sort == sort by (score desc, anchor index asc),
// IoU in f64 on f32 coordinates with strict `> 0.3`, weighted sums accumulated in sorted order in f32 with separate
// multiply and add (no FMA), output score = head score, stop when nothing was removed.

__device__ __forceinline__ float sigmoid_f32(float x) { return __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x))); }

// feature f (0..15) of the decoded detection of anchor `a` (face_detection.rs:274-293)
__device__ __forceinline__ float decode_feature(const float* __restrict__ rb, const float* __restrict__ anchors, int a, int f,
                                                float scale) {
    const float* r = rb + (long)a * 16;
    float anc = anchors[2 * a + (f & 1)];
    if (f < 4) {
        float c = __fadd_rn(__fdiv_rn(r[f & 1], scale), anc);            // centre + anchor
        float h = __fdiv_rn(__fdiv_rn(r[2 + (f & 1)], scale), 2.0f);     // size / 2 (no anchor term)
        return f < 2 ? __fsub_rn(c, h) : __fadd_rn(c, h);
    }
    return __fadd_rn(__fdiv_rn(r[f], scale), anc);
}

__device__ __forceinline__ double iou_f64(const float* b1, const float* b2) {  // nms.rs:5-17, types.rs:123-159
    double x0 = fmax((double)b1[0], (double)b2[0]), y0 = fmax((double)b1[1], (double)b2[1]);
    double x1 = fmin((double)b1[2], (double)b2[2]), y1 = fmin((double)b1[3], (double)b2[3]);
    if (!(x0 < x1 && y0 < y1)) return 0.0;
    double iw = x1 - x0, ih = y1 - y0;
    double ia = (iw <= 0.0 || ih <= 0.0) ? 0.0 : iw * ih;
    double w1 = (double)b1[2] - (double)b1[0], h1 = (double)b1[3] - (double)b1[1];
    double w2 = (double)b2[2] - (double)b2[0], h2 = (double)b2[3] - (double)b2[1];
    double a1 = (w1 <= 0.0 || h1 <= 0.0) ? 0.0 : w1 * h1;
    double a2 = (w2 <= 0.0 || h2 <= 0.0) ? 0.0 : w2 * h2;
    double den = a1 + a2 - ia;
    return den > 0.0 ? ia / den : 0.0;
}

template <int NP>  // NP = power of two >= number of anchors
__global__ __launch_bounds__(256) void ssd_postprocess_kernel(PostArgs a) {
    __shared__ unsigned long long keys[NP];   // (~score bits) << 32 | anchor index; sorted ascending
    __shared__ float boxes[NP][4];            // decoded (xmin,ymin,xmax,ymax) per sorted position (first M valid)
    __shared__ unsigned char state[NP];       // 1 = still in `remaining`, 2 = candidate of the current head
    __shared__ int s_count, s_removed, s_head;
    __shared__ float s_w[17];
    const int tid = threadIdx.x, b = blockIdx.x;
    const float* rb = a.raw_boxes + (long)b * a.N * 16;
    const float* rs = a.raw_scores + (long)b * a.N;
    if (tid == 0) s_count = 0;
    __syncthreads();
    // --- threshold + validity (convert_to_detections); order-insensitive compaction, the sort restores the order
    for (int i = tid; i < a.N; i += 256) {
        float x = rs[i];
        x = x < -80.0f ? -80.0f : (x > 80.0f ? 80.0f : x);
        float s = sigmoid_f32(x);
        if (s > 0.5f) {
            float x0 = decode_feature(rb, a.anchors, i, 0, a.scale), y0 = decode_feature(rb, a.anchors, i, 1, a.scale);
            float x1 = decode_feature(rb, a.anchors, i, 2, a.scale), y1 = decode_feature(rb, a.anchors, i, 3, a.scale);
            if (x1 > x0 && y1 > y0) {
                int slot = atomicAdd(&s_count, 1);
                keys[slot] = ((unsigned long long)(0xFFFFFFFFu - __float_as_uint(s)) << 32) | (unsigned)i;
            }
        }
    }
    __syncthreads();
    const int M = s_count;
    int P = 1;
    while (P < M) P <<= 1;
    for (int i = M + tid; i < P; i += 256) keys[i] = ~0ull;
    __syncthreads();
    // --- bitonic sort of P keys (unique keys => deterministic, equals the reference's stable sort)
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += 256) {
                int l = i ^ j;
                if (l > i) {
                    unsigned long long x = keys[i], y = keys[l];
                    bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[l] = x; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < M; i += 256) {
        int an = (int)(keys[i] & 0xFFFFFFFFu);
#pragma unroll
        for (int f = 0; f < 4; f++) boxes[i][f] = decode_feature(rb, a.anchors, an, f, a.scale);
        state[i] = 1;
    }
    if (tid == 0) s_head = 0;
    __syncthreads();
    // letterbox constants (transform.rs:116-141)
    float lb_l = 0.f, lb_t = 0.f, lb_hs = 1.f, lb_vs = 1.f;
    bool lb_bad = false;
    if (a.padding) {
        const double* pd = a.padding + (long)b * 4;
        double hs = 1.0 - (pd[0] + pd[2]), vs = 1.0 - (pd[1] + pd[3]);
        lb_bad = !(hs > 2.220446049250313e-16) || !(vs > 2.220446049250313e-16);
        lb_l = (float)pd[0]; lb_t = (float)pd[1]; lb_hs = (float)hs; lb_vs = (float)vs;
    }
    int nout = 0;
    const double thr = (double)0.3f;  // MIN_SUPPRESSION_THRESHOLD as f64 (face_detection.rs:139, nms.rs:87)
    float* outp = a.out + (long)b * a.cap * 17;
    while (true) {
        // head = first entry of `remaining`
        if (tid == 0) {
            int h = s_head;
            while (h < M && state[h] == 0) h++;
            s_head = h;
            s_removed = 0;
        }
        __syncthreads();
        const int head = s_head;
        if (head >= M) break;
        const float head_score = __uint_as_float(0xFFFFFFFFu - (unsigned)(keys[head] >> 32));
        // (head.score < MIN_SCORE cannot happen: every kept score is > 0.5 — nms.rs:69-73)
        int removed = 0;
        for (int i = head + tid; i < M; i += 256) {
            if (state[i] == 0) continue;
            double sim = iou_f64(boxes[i], boxes[head]);
            if (sim > thr) { state[i] = 2; removed++; }
        }
        if (removed) atomicAdd(&s_removed, removed);
        __syncthreads();
        const int nrem = s_removed;
        // weighted merge (nms.rs:94-112): lanes 0..15 own one feature each, lane 16 owns total_score; sequential
        // f32 accumulation in sorted order, separate multiply and add.
        if (tid < 17) {
            float acc = 0.0f;
            if (nrem > 0) {
                for (int i = head; i < M; i++) {
                    if (state[i] != 2) continue;
                    unsigned long long k = keys[i];
                    float s = __uint_as_float(0xFFFFFFFFu - (unsigned)(k >> 32));
                    if (tid == 16) acc = __fadd_rn(acc, s);
                    else acc = __fadd_rn(acc, __fmul_rn(decode_feature(rb, a.anchors, (int)(k & 0xFFFFFFFFu), tid, a.scale), s));
                }
            } else if (tid < 16) {
                acc = decode_feature(rb, a.anchors, (int)(keys[head] & 0xFFFFFFFFu), tid, a.scale);  // detection.clone()
            }
            s_w[tid] = acc;
        }
        __syncthreads();
        if (tid < 17 && nout < a.cap) {
            float v;
            if (tid == 16) v = head_score;
            else {
                v = nrem > 0 ? __fdiv_rn(s_w[tid], s_w[16]) : s_w[tid];
                if (a.padding)  // detection_letterbox_removal: (v - left) / h_scale, (v - top) / v_scale
                    v = (tid & 1) ? __fdiv_rn(__fsub_rn(v, lb_t), lb_vs) : __fdiv_rn(__fsub_rn(v, lb_l), lb_hs);
            }
            outp[(long)nout * 17 + tid] = v;
        }
        nout++;
        for (int i = head + tid; i < M; i += 256)
            if (state[i] == 2) state[i] = 0;
        __syncthreads();
        if (nrem == 0) break;  // "number of indexed scores didn't change" (nms.rs:117-119)
    }
    if (tid == 0) a.counts[b] = lb_bad ? -1 : nout;  // -1: the reference's letterbox assert! would have fired
    if (a.zero_rest)
        for (int i = min(nout, a.cap) * 17 + tid; i < a.cap * 17; i += 256) outp[i] = 0.f;
    if (a.face_rois && tid == 0) {   // (slot 0 was written by this workgroup, barriers since)
        RectD r;
        a.face_valid[b] = face_roi_dev(outp, lb_bad ? -1 : nout, a.image_w, a.image_h, &r);
        a.face_rois[b] = r;
    }
}

int launch_postprocess(const PostArgs& a, void* stream) {
    if (a.B <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (a.N <= 1024) hipLaunchKernelGGL(ssd_postprocess_kernel<1024>, dim3(a.B), dim3(256), 0, s, a);
    else if (a.N <= 2048) hipLaunchKernelGGL(ssd_postprocess_kernel<2048>, dim3(a.B), dim3(256), 0, s, a);
    else if (a.N <= 4096) hipLaunchKernelGGL(ssd_postprocess_kernel<4096>, dim3(a.B), dim3(256), 0, s, a);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ landmark projection
// project_landmarks (transform.rs:351-432); one thread per landmark; f32/f64 mix kept as in the Rust source.
__global__ void project_landmarks_kernel(ProjArgs a) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nt = a.n + a.n2;
    if (idx >= (long)a.B * nt) return;
    const int b = (int)(idx / nt), k = (int)(idx % nt);
    // (a second list of landmarks of the same items — the iris network's two outputs, iris_landmark.rs:213-245, in one launch)
    const bool second = k >= a.n;
    const int li = second ? k - a.n : k;            // index within its list
    const int i = second ? -1 : li;                 // i == 0: the item's first landmark (the flag and the gate bookkeeping belong to it)
    float* o = (second ? a.out2 + (long)b * (a.out2_fs ? a.out2_fs : 3L * a.n2) : a.out + (long)b * (a.out_fs ? a.out_fs : 3L * a.n)) + 3 * li;
    if (a.gate && !a.gate[b]) {
        o[0] = o[1] = o[2] = 0.f;
        if (i == 0 && a.present) a.present[b] = 0;
        if (i == 0 && a.raw_flag_out && a.flag) a.raw_flag_out[b] = a.flag[(long)b * a.flag_fs];
        return;
    }
    const float* r = (second ? a.raw2 + (long)b * a.raw2_fs : a.raw + (long)b * a.raw_fs) + 3 * li;
    float wf = (float)a.tensor_w, hf = (float)a.tensor_h;
    float x = __fdiv_rn(r[0], wf), y = __fdiv_rn(r[1], hf), z = __fdiv_rn(r[2], wf);
    if (a.flip && a.flip[b]) x = __fadd_rn(__fmul_rn(x, -1.0f), 1.0f);
    if (a.padding) {
        const double* pd = a.padding + (long)b * 4;
        if (!(pd[0] == 0.0 && pd[1] == 0.0 && pd[2] == 0.0 && pd[3] == 0.0)) {
            double hs = 1.0 - (pd[0] + pd[2]), vs = 1.0 - (pd[1] + pd[3]);
            x = (float)(((double)x - pd[0]) / hs);
            y = (float)(((double)y - pd[1]) / vs);
            z = (float)(((double)z - 0.) / hs);
        }
    }
    if (i == 0 && a.flag) {
        float fl = a.flag[(long)b * a.flag_fs];
        if (a.present) a.present[b] = sigmoid_f32(fl) <= 0.5f ? 0 : 1;
        if (a.raw_flag_out) a.raw_flag_out[b] = fl;
    }
    if (a.roi) {
        const RectD ro = a.roi[b];
        double xc = ro.x_center, yc = ro.y_center, w = ro.width, h = ro.height, rot = ro.rotation;
        if (!ro.normalized) {  // Rect::scaled(size, true): multiply by the reciprocal (types.rs:67)
            double sx = 1.0 / (double)a.image_size[2 * b], sy = 1.0 / (double)a.image_size[2 * b + 1];
            xc *= sx; yc *= sy; w *= sx; h *= sy;
        }
        float m00 = (float)cos(rot), m01 = (float)sin(rot), m10 = (float)(-sin(rot)), m11 = m00;
        x = __fsub_rn(x, 0.5f); y = __fsub_rn(y, 0.5f); z = __fsub_rn(z, 0.0f);
        float rz = __fmul_rn(z, 0.0f);
        float rx = __fadd_rn(__fadd_rn(__fmul_rn(x, m00), __fmul_rn(y, m10)), __fmul_rn(rz, 1.0f));
        float ry = __fadd_rn(__fadd_rn(__fmul_rn(x, m01), __fmul_rn(y, m11)), __fmul_rn(rz, 1.0f));
        float rzz = __fadd_rn(__fadd_rn(__fmul_rn(x, 0.0f), __fmul_rn(y, 0.0f)), __fmul_rn(rz, 1.0f));
        x = __fadd_rn(__fmul_rn(x, 0.f), rx);
        y = __fadd_rn(__fmul_rn(y, 0.f), ry);
        z = __fadd_rn(__fmul_rn(z, 1.f), rzz);
        x = (float)((double)x * w + xc);
        y = (float)((double)y * h + yc);
        z = (float)((double)z * w + 0.);
    }
    o[0] = x; o[1] = y; o[2] = z;
}

int launch_project(const ProjArgs& a, void* stream) {
    long n = (long)a.B * (a.n + a.n2);
    if (n > 0) hipLaunchKernelGGL(project_landmarks_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// One wave that does nothing for `ticks` of the 100 MHz wall clock (bounded: it leaves after `max_iters` looks at the clock whatever the
// clock says).  mi_streams_create_distinct launches it on several streams at once: streams that share a hardware queue run it one after
// the other, streams on different queues side by side.
__global__ void spin_kernel(long long ticks, int max_iters, int* sink) {
    const long long t0 = wall_clock64();
    int it = 0;
    while (wall_clock64() - t0 < ticks && ++it < max_iters) __builtin_amdgcn_s_sleep(8);
    if (ticks < 0) *sink = it;
}

int launch_spin(long long ticks, int max_iters, int* sink, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks, max_iters, sink);
    return (int)hipGetLastError();
}

}  // namespace mi
