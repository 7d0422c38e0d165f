// strip_kernels.hip — register-resident fused BlazeBlock for the narrow, stride-1 layers (C = Co in {16, 24, 32}).
//
//   out = act( PW1x1( DW3x3(in) + b_dw ) + b_pw + in )
//
// Same operator chain as block_kernels.hip (DEPTHWISE_CONV_2D -> CONV_2D 1x1 -> ADD -> RELU/PRELU behind
// `interpreter.invoke()`, /root/reference/src/face_detection_lite/face_detection.rs:235; the 24-channel blocks at 128^2
// and 64^2 are 14 of BackCamera's 37 launches and two thirds of its time), different mapping to the machine.  The
// block kernel keeps rows in a workgroup-shared LDS ring and re-reads the 3x3 window and the depthwise weights from
// LDS per lane: ~100 ds_read_b128 per 64 pixels, which keeps the CU's LDS pipe ~60 % busy at 2 waves/SIMD — its wall.
// Here:
//   * one WAVE owns a strip of 64 pixel columns and walks down a band of rows on its own: no workgroup barrier at all.
//   * lane = pixel.  A row of the strip (64 + 2 halo pixels, all channels) is fetched with fully coalesced 16 B/lane
//     loads (1 KiB contiguous per instruction), transposed through a small wave-private LDS buffer (pixel stride C+4
//     floats: conflict-free b128 accesses), after which the lane holds its pixel's channels and reads its left / right
//     neighbours from the same buffer: 3 ds_read_b128 per channel quad and row — each input element is read from LDS
//     three times instead of nine, and vertical reuse is in registers: the three partial output rows a new input row
//     contributes to (ky = 2, 1, 0) stay in VGPR accumulators.
//   * depthwise weights / biases are wave-uniform in this layout: they arrive through the scalar cache as SGPR-pair
//     operands of v_pk_fma_f32 (two channels per VALU lane-op), no LDS or VGPR cost.
//   * the finished depthwise row (lane = pixel) becomes the B operand of v_mfma_f32_32x32x2_f32 (B[k = lane>>5][n =
//     lane&31]) with one v_permlane32_swap per channel pair: swap(ch 2j, ch 2j+1) yields the operands of both 32-pixel
//     halves.  The pointwise weights sit in C/2 VGPRs for the whole kernel (A[i = lane&31][k = lane>>5]).
//   * D comes back to lane = pixel with one swap per register pair; + bias (SGPR) + skip (the centre input row, still
//     in VGPRs) -> activation -> transposed back through LDS so that every store instruction writes 1 KiB of
//     consecutive bytes (full 128-byte lines instead of 16-byte pieces that have to meet in L2).
//   * the next input row is loaded into registers before the current row's math and staged to LDS after it.
// Exact f32 throughout (MFMA f32 = fmaf chain); results match the block kernel to reassociation of the 3x3 sum.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"

namespace mi {

namespace {

typedef float sf32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float __attribute__((address_space(4))) cfloat;

struct StripArgs {
    const float* in;
    float* out;
    const float* consts;   // strip_pack_consts() blob
    long in_fs, out_fs;
    int B, H, W;
    int strips, bands, band_rows;
    int has_res;
    float hi;              // upper clamp of the activation (6 for ReLU6, +inf otherwise)
};

template <int CQ>
struct SK {
    static constexpr int C = 4 * CQ;          // channels
    static constexpr int PS = C + 4;          // LDS pixel stride (floats)
    static constexpr int K2 = C / 2;          // MFMA k-steps (2 channels each)
    static constexpr int NF = 66 * CQ;        // float4s of one staged input row (64 pixels + 2 halo pixels)
    static constexpr int NL = (NF + 63) / 64; // float4s per lane of that row
    static constexpr int IN_F = 66 * PS, OUT_F = 64 * PS, WAVE_F = IN_F + OUT_F;
    // constants blob (floats): per channel pair 9 taps x 2 (+2 spare); pointwise bias (+ W b_dw); negative slopes; A fragments
    static constexpr int OFF_DW = 0, OFF_BIAS = CQ * 40, OFF_SLOPE = OFF_BIAS + 32, OFF_A = OFF_SLOPE + 32, TOTAL = OFF_A + K2 * 64;
};

__device__ __forceinline__ float4 sld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void sst4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// LDS traffic between the lanes of ONE wave: the hardware executes a wave's LDS instructions in order; this only has to
// stop the compiler from moving accesses across the hand-over.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ v2f pkfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

template <int CQ, bool RELU>
__global__ __launch_bounds__(256, 2) void strip_kernel(StripArgs a) {
    using K = SK<CQ>;
    constexpr int C = K::C, PS = K::PS, NL = K::NL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* inbuf = lds + wave * K::WAVE_F;
    float* outbuf = inbuf + K::IN_F;
    int gw = blockIdx.x * 4 + wave;
    const int band = gw % a.bands;
    gw /= a.bands;
    const int strip = gw % a.strips, b = gw / a.strips;
    if (b >= a.B) return;  // whole wave; no workgroup-level synchronisation anywhere in this kernel
    const int x0 = strip * 64;
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, a.H);
    if (y0 >= y1) return;
    const float* in = a.in + (long)b * a.in_fs;
    float* out = a.out + (long)b * a.out_fs;
    // constant address space: the blob is never written while kernels run, and wave-uniform loads from it become s_load
    const cfloat* cst = (const cfloat*)a.consts;

    float A[K::K2];
#pragma unroll
    for (int j = 0; j < K::K2; j++) A[j] = cst[K::OFF_A + j * 64 + lane];

    // transpose geometry: float4 number f = lane + 64k of a staged row <-> (pixel f / CQ, channel quad f % CQ).
    // Loads are unconditional: a lane whose pixel lies outside the image (the halo of an edge strip, the tail of a
    // partial strip) reads a clamped in-image pixel instead, and the one staged pixel column that must be zero (left of
    // x = 0, right of x = W-1) is cleared in LDS after staging — no per-lane branches around the memory instructions.
    int loff[NL], goff[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int f = min(lane + 64 * k, K::NF - 1), px = f / CQ, qd = f - px * CQ;
        loff[k] = px * PS + 4 * qd;
        goff[k] = min(max(x0 - 1 + px, 0), a.W - 1) * C + 4 * qd;
    }
    const int npx = min(64, a.W - x0);              // valid pixels of this strip
    const bool full = npx == 64;
    const int zl = x0 == 0 ? 0 : -1;                // staged pixel columns to clear (-1: none)
    const int zr = x0 + 64 >= a.W ? npx + 1 : -1;
    const long gout = (long)x0 * C + 4 * lane;

    float4 pf[NL];
    auto load_row = [&](int r) {
        if (r >= 0 && r < a.H) {  // wave-uniform
            const float* src = in + (long)r * a.W * C;
#pragma unroll
            for (int k = 0; k < NL; k++) pf[k] = sld4(src + goff[k]);
        } else {
#pragma unroll
            for (int k = 0; k < NL; k++) pf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage_row = [&]() {
#pragma unroll
        for (int k = 0; k < NL; k++)
            if (64 * (k + 1) <= K::NF || lane + 64 * k < K::NF) sst4(inbuf + loff[k], pf[k]);
        if (lane < CQ) {
            if (zl >= 0) sst4(inbuf + zl * PS + 4 * lane, make_float4(0.f, 0.f, 0.f, 0.f));
            if (zr >= 0) sst4(inbuf + zr * PS + 4 * lane, make_float4(0.f, 0.f, 0.f, 0.f));
        }
    };

    // One input row r (staged in inbuf): its ky = 2 / 1 / 0 taps go to the partial output rows r-1 / r / r+1 (aP / aC /
    // aN).  Work is cut into 2*CQ stages of one channel PAIR each (= one MFMA k-step): stage s waits for its own
    // operands (20 scalar weights, 3 float4 LDS reads per quad), issues the loads of stage s+1, then finishes the pair of
    // row r-1 (3 v_pk_fma), swaps it into the two B operands and issues the two MFMAs; the remaining 6 v_pk_fma of the
    // pair (rows r, r+1) and the next stage's load latency sit in the shadow of those 128 MFMA-pipe cycles.
    sf32x16 D0, D1;
    auto row = [&](auto emit_t, v2f (&aP)[CQ][2], v2f (&aC)[CQ][2], v2f (&aN)[CQ][2], float4 (&xcur)[CQ]) {
        constexpr bool EMIT = decltype(emit_t)::value;
        const float* me = inbuf + lane * PS;
#pragma unroll
        for (int e = 0; e < 16; e++) { D0[e] = 0.f; D1[e] = 0.f; }
        float wbuf[2][20];
        float4 xbuf[2][3];
        auto load_w = [&](int st, float (&w)[20]) {
            const cfloat* wp = cst + K::OFF_DW + st * 20;
            asm volatile("" : "+s"(wp));  // opaque per use: the scalar loads stay in the row loop, one stage ahead of their use
#pragma unroll
            for (int i = 0; i < 20; i++) w[i] = wp[i];
        };
        auto load_x = [&](int q, float4 (&x)[3]) {
#pragma unroll
            for (int kx = 0; kx < 3; kx++) x[kx] = sld4(me + kx * PS + 4 * q);
        };
        load_w(0, wbuf[0]);
        load_x(0, xbuf[0]);
#pragma unroll
        for (int st = 0; st < 2 * CQ; st++) {
            const int q = st >> 1, h = st & 1;
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this stage's weights and pixels have landed ...
            if (st + 1 < 2 * CQ) load_w(st + 1, wbuf[(st + 1) & 1]);  // ... so the next stage's loads never delay this stage's math
            if (h == 0 && q + 1 < CQ) load_x(q + 1, xbuf[(q + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const float (&w)[20] = wbuf[st & 1];
            const float4 (&x)[3] = xbuf[q & 1];
            if (h == 0) xcur[q] = x[1];
            v2f t[3];
#pragma unroll
            for (int kx = 0; kx < 3; kx++) t[kx] = h == 0 ? v2f{x[kx].x, x[kx].y} : v2f{x[kx].z, x[kx].w};
            auto wv = [&](int tap) { return v2f{w[2 * tap], w[2 * tap + 1]}; };
            if constexpr (EMIT) {
                v2f pch = aP[q][h];
#pragma unroll
                for (int kx = 0; kx < 3; kx++) pch = pkfma(t[kx], wv(6 + kx), pch);
                const v2u sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(pch.x), __float_as_uint(pch.y), false, false);
                D0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[st], __uint_as_float(sw.x), D0, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[st], __uint_as_float(sw.y), D1, 0, 0, 0);
            }
            v2f n = t[0] * wv(0), c = aC[q][h];  // the depthwise bias is folded into the pointwise bias on the host
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                if (kx) n = pkfma(t[kx], wv(kx), n);
                c = pkfma(t[kx], wv(3 + kx), c);
            }
            aC[q][h] = c;
            aN[q][h] = n;
            // pin the partial rows here: left alone, LLVM sinks these updates past the loop-exit test into the next row's
            // block, across the scheduling fences, which keeps every stage's weights alive (SGPR spills)
            asm volatile("" : "+v"(aC[q][h]), "+v"(aN[q][h]));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // epilogue of output row y: D back to lane = pixel, + bias + skip, activation, transposed store
    auto epilogue = [&](int y, const float4 (&xres)[CQ]) {
#pragma unroll
        for (int g = 0; g < CQ / 2; g++) {
            float4 v[2];
            float* vv = reinterpret_cast<float*>(v);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const v2u s = __builtin_amdgcn_permlane32_swap(__float_as_uint(D0[4 * g + e]), __float_as_uint(D1[4 * g + e]), false, false);
                vv[e] = __uint_as_float(s.x);      // channel 8g + e     of pixel `lane`
                vv[4 + e] = __uint_as_float(s.y);  // channel 8g + 4 + e
            }
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int q = 2 * g + hh;
                const cfloat* bq = cst + K::OFF_BIAS + 4 * q;
                asm volatile("" : "+s"(bq));
                float4 o = make_float4(v[hh].x + bq[0], v[hh].y + bq[1], v[hh].z + bq[2], v[hh].w + bq[3]);
                if (a.has_res) { o.x += xres[q].x; o.y += xres[q].y; o.z += xres[q].z; o.w += xres[q].w; }
                if (RELU) {
                    o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                } else {
                    // act(v) = min(max(v,0) + slope*min(v,0), hi): ReLU (slope 0), PReLU (alpha), none (1), ReLU6 (hi = 6)
                    const cfloat* sq = cst + K::OFF_SLOPE + 4 * q;
                    o.x = fminf(fmaxf(o.x, 0.f) + sq[0] * fminf(o.x, 0.f), a.hi);
                    o.y = fminf(fmaxf(o.y, 0.f) + sq[1] * fminf(o.y, 0.f), a.hi);
                    o.z = fminf(fmaxf(o.z, 0.f) + sq[2] * fminf(o.z, 0.f), a.hi);
                    o.w = fminf(fmaxf(o.w, 0.f) + sq[3] * fminf(o.w, 0.f), a.hi);
                }
                sst4(outbuf + lane * PS + 4 * q, o);
            }
        }
        wave_sync();
        float* dst = out + (long)y * a.W * C + gout;
        if (full) {  // wave-uniform
#pragma unroll
            for (int k = 0; k < CQ; k++) sst4(dst + 256 * k, sld4(outbuf + loff[k]));
        } else {
#pragma unroll
            for (int k = 0; k < CQ; k++) {
                const float4 o = sld4(outbuf + loff[k]);
                if (lane + 64 * k < npx * CQ) sst4(dst + 256 * k, o);
            }
        }
    };

    v2f acc0[CQ][2], acc1[CQ][2], acc2[CQ][2];
    float4 xa[CQ], xb[CQ], xd[CQ];
#pragma unroll
    for (int q = 0; q < CQ; q++) {
        acc0[q][0] = acc0[q][1] = acc1[q][0] = acc1[q][1] = acc2[q][0] = acc2[q][1] = v2f{0.f, 0.f};
        xa[q] = xb[q] = xd[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    load_row(y0 - 1);
    stage_row();
    wave_sync();
    // EMIT is a compile-time property of the call site (the two priming rows of a band produce no output): as a run-time
    // branch the compiler sinks a third of the depthwise FMAs into it, across the scheduling fences, and spills SGPRs.
    auto step = [&](auto emit, int r, v2f (&aP)[CQ][2], v2f (&aC)[CQ][2], v2f (&aN)[CQ][2], const float4 (&xprev)[CQ], float4 (&xcur)[CQ]) {
        const bool more = r < y1;  // input rows run to y1 inclusive
        if (more) load_row(r + 1);
        row(emit, aP, aC, aN, xcur);
        wave_sync();  // every read of inbuf above is issued before the writes below
        if (more) stage_row();  // before the epilogue: its stores are then never waited on until a row later, and the
                                // staging covers the drain of the last MFMAs
        if constexpr (decltype(emit)::value) epilogue(r - 1, xprev);
        wave_sync();
    };
    step(std::false_type{}, y0 - 1, acc0, acc1, acc2, xd, xa);
    step(std::false_type{}, y0, acc1, acc2, acc0, xa, xb);
    for (int r = y0 + 1; r <= y1; r += 3) {
        step(std::true_type{}, r, acc2, acc0, acc1, xb, xd);
        if (r + 1 > y1) break;
        step(std::true_type{}, r + 1, acc0, acc1, acc2, xd, xa);
        if (r + 2 > y1) break;
        step(std::true_type{}, r + 2, acc1, acc2, acc0, xa, xb);
    }
}

int strips_band_rows(const BlockArgs& a, int strips) {
    // about one resident set of waves over the chip (8 per CU), but bands of at least 8 rows (2 halo rows re-read per band)
    static const int forced = getenv("MI_STRIP_BAND") ? atoi(getenv("MI_STRIP_BAND")) : 0;  // tuning aid
    if (forced > 0) return std::min(forced, a.H);
    const long per_row_waves = (long)a.B * strips;
    long bands = std::max<long>(1, (2048 + per_row_waves / 2) / per_row_waves);
    int rows = (int)((a.H + bands - 1) / bands);
    rows = std::max(rows, std::min(a.H, 8));
    return rows;
}

template <int CQ, bool RELU>
int launch_strip_inst(const BlockArgs& a, hipStream_t s) {
    using K = SK<CQ>;
    auto kern = strip_kernel<CQ, RELU>;
    StripArgs sa;
    sa.in = a.in; sa.out = a.out; sa.consts = a.w_strip; sa.in_fs = a.in_fs; sa.out_fs = a.out_fs;
    sa.B = a.B; sa.H = a.H; sa.W = a.W;
    sa.strips = (a.W + 63) / 64;
    sa.band_rows = strips_band_rows(a, sa.strips);
    sa.bands = (a.H + sa.band_rows - 1) / sa.band_rows;
    sa.has_res = a.ep.res_mode == RES_DIRECT;
    sa.hi = a.ep.act == ACT_RELU6 ? 6.f : INFINITY;
    const long waves = (long)a.B * sa.strips * sa.bands;
    const size_t lds_bytes = (size_t)4 * K::WAVE_F * 4;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((waves + 3) / 4)), dim3(256), lds_bytes, s, sa);
    return (int)hipGetLastError();
}

}  // namespace

// Shapes the strip kernel takes: stride-1 depthwise 3x3 (SAME) + pointwise with C = Co in {16, 24, 32}, skip = the
// block's own input (or none), constants packed by strip_pack_consts().
bool strip_kernel_supports(const BlockArgs& a) {
    static const bool off = getenv("MI_NO_STRIP") != nullptr;  // tuning aid: fall back to the LDS-ring block kernel
    if (off || !a.w_strip || !a.has_dw || a.sh != 1 || a.sw != 1 || a.pt != 1 || a.pl != 1) return false;
    if (a.C != a.Co || (a.C != 16 && a.C != 24 && a.C != 32) || a.H != a.Ho || a.W != a.Wo) return false;
    if (a.ep.res_mode != RES_NONE) {
        if (a.ep.res_mode != RES_DIRECT || a.ep.res != a.in || a.ep.res_fs != a.in_fs || a.ep.res_C != a.C) return false;
    }
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!aligned16(a.in) || !aligned16(a.out) || (a.in_fs & 3) || (a.out_fs & 3)) return false;
    return true;
}

bool strip_shape_ok(int C, int Co) { return C == Co && (C == 16 || C == 24 || C == 32); }

int strip_consts_floats(int C) { return C / 4 * 40 + 64 + C / 2 * 64; }

// w_dw [3][3][C], b_dw [C] or null, w_pw [Co][C] (TFLite OHWI with H = W = 1), bias [Co] or null, alpha [Co] or null.
void strip_pack_consts(int C, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst) {
    const int CQ = C / 4;
    std::fill(dst, dst + strip_consts_floats(C), 0.f);
    for (int st = 0; st < 2 * CQ; st++)  // channel pair (2st, 2st+1): 9 taps x 2 (the last 2 floats of the 20 are spare)
        for (int t = 0; t < 9; t++)
            for (int e = 0; e < 2; e++) dst[st * 20 + 2 * t + e] = w_dw[t * C + 2 * st + e];
    float* pb = dst + CQ * 40;
    float* ps = pb + 32;
    float* pa = ps + 32;
    for (int c = 0; c < C; c++) {
        // PW(dw + b_dw) + b_pw = PW(dw) + (W b_dw + b_pw): the depthwise bias is folded into the pointwise bias
        double acc = bias ? bias[c] : 0.0;
        if (b_dw)
            for (int k = 0; k < C; k++) acc += (double)w_pw[(size_t)c * C + k] * b_dw[k];
        pb[c] = (float)acc;
        ps[c] = act == ACT_PRELU ? alpha[c] : (act == ACT_NONE ? 1.f : 0.f);
    }
    // A fragment of k-step j: lane l holds W[out = l & 31][in = 2j + (l >> 5)]
    for (int j = 0; j < C / 2; j++)
        for (int l = 0; l < 64; l++) {
            const int o = l & 31, c = 2 * j + (l >> 5);
            pa[j * 64 + l] = o < C ? w_pw[(size_t)o * C + c] : 0.f;
        }
}

const char* strip_kernel_label(const BlockArgs& a, char* buf, size_t cap) {
    snprintf(buf, cap, "strip_kernel<%d,%d>", a.C / 4, a.ep.act == ACT_RELU ? 1 : 0);
    return buf;
}

int launch_strip(const BlockArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const bool relu = a.ep.act == ACT_RELU;
#define MI_STRIP_CASE(Q) \
    if (a.C == 4 * Q) return relu ? launch_strip_inst<Q, true>(a, s) : launch_strip_inst<Q, false>(a, s);
    MI_STRIP_CASE(4) MI_STRIP_CASE(6) MI_STRIP_CASE(8)
#undef MI_STRIP_CASE
    return (int)hipErrorInvalidValue;
}

}  // namespace mi
