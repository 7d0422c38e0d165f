// ms2_kernels.hip — the stride-2 BlazeBlock in the matrix operand layout (the mstrip / mdblock scheme, mrow.hpp's conventions):
//
//   out = act( PW1x1( DW3x3_s2(in) + b_dw ) + b_pw + pad_channels( maxpool2x2(in) ) )        (the skip is optional)
//
// (DEPTHWISE_CONV_2D stride 2 -> CONV_2D 1x1 [-> MAX_POOL_2D, PAD, ADD] -> PRELU / RELU behind `interpreter.invoke()`,
// /root/reference/src/face_detection_lite/face_landmark.rs:265 and face_detection.rs:235; SURVEY.md Appendix A.2 / A.3: the face mesh's
// 48x48x32 -> 24x24x64, full_range's 96x96x32 -> 48x48x12 and 48x48x64 -> 24x24x24.)  The block kernel runs these at 1.5 - 2.5x their
// traffic time (0.096 ms for the mesh block at 512 ROIs: 226 MB).
//
// A workgroup = NF frames x NWV waves on one band of OUTPUT rows, each wave owning one 16-pixel output tile (32 input pixels of every
// input row), all walking down in step; a step brings TWO input rows (2 s, 2 s + 1) by LDS-DMA into shared row images:
//   row 2 s     finishes output row s - 1 (its ky = 2 taps; MFMAs, epilogue) and starts output row s (ky = 0),
//   row 2 s + 1 adds the ky = 1 taps; then output row s's accumulators start from bias + the 2 x 2 max-pool of the two rows.
// Lane (kq = lane / 16, p = lane % 16) owns channel 4 ks + kq of output pixel p: it reads input pixels 2 p, 2 p + 1, 2 p + 2 (TFLite SAME
// for even sizes pads right / bottom only: column W_in and row H_in are zeros).  Four row images per frame: the rows of step s + 2 are
// asked for when step s has released its two.  Two workgroup barriers per step (rows complete; rows released).
// Exact f32; results match the block kernel to reassociation of the sums and the folded depthwise bias.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "kernels.hpp"
#include "launch.hpp"
#include "mrow.hpp"

namespace mi {

namespace {

struct Ms2Args {
    const float* in;
    float* out;
    const float* consts;   // ms2_pack_consts()
    long in_fs, out_fs;
    int B, Ho;             // frames, output rows (input: 2 Ho rows)
    int bands, band_rows;
    float hi;
};

// CK = C / 4 k-steps, MT output tiles (16 MT >= CO real output channels), NWV waves = output tiles per row (WO <= 16 NWV output pixels),
// NF frames per workgroup, SKIP: max-pool skip, TAPL: taps from LDS per k-step
template <int CK_, int MT_, int CO_, int NWV_, int WO_, int NF_, bool SKIP_, bool TAPL_>
struct MS2 {
    static constexpr int CK = CK_, MT = MT_, CO = CO_, NWV = NWV_, WO = WO_, NF = NF_, C = 4 * CK, WI = 2 * WO;
    static constexpr bool SKIP = SKIP_, TAPL = TAPL_;
    static_assert(WO <= 16 * NWV && WO > 16 * (NWV - 1) && CO <= 16 * MT && CO % 4 == 0, "tiles cover the row / the channels");
    static constexpr int QP = CK + 1, PS = 4 * QP;
    static_assert(QP % 2 == 1, "odd number of float4 slots per pixel");
    static constexpr int IMG_F = (WI + 1) * PS;                 // input pixels 0 .. WI - 1 and the zero column WI
    static constexpr int DPX = QP <= 6 ? 8 : (QP <= 16 ? 4 : 2), ACTIVE = DPX * QP;   // LDS-DMA: pixels per instruction (more than half a wave of lanes); four instructions (one source base) per group
    static_assert(ACTIVE > 32 && ACTIVE <= 64 && 3 * DPX * C * 4 < 4096, "DMA shape");
    static constexpr int OFF_A = 0, A_F = CK * MT * 64, OFF_TAP = A_F, TAP_F = CK * 48, OFF_BIAS = OFF_TAP + TAP_F, OFF_SLOPE = OFF_BIAS + 16 * MT, TOTAL = OFF_SLOPE + 16 * MT;
    static constexpr int LDS_F = TOTAL + NF * 4 * IMG_F;
};

__device__ __forceinline__ void ms2_wait_le(int n) {   // s_waitcnt vmcnt(n), n a multiple of 4 up to 44 (wave-uniform)
    switch (n >> 2) {
        case 0: dwait_vm<0>(); break;  case 1: dwait_vm<4>(); break;  case 2: dwait_vm<8>(); break;   case 3: dwait_vm<12>(); break;
        case 4: dwait_vm<16>(); break; case 5: dwait_vm<20>(); break; case 6: dwait_vm<24>(); break;  case 7: dwait_vm<28>(); break;
        case 8: dwait_vm<32>(); break; case 9: dwait_vm<36>(); break; case 10: dwait_vm<40>(); break; default: dwait_vm<44>(); break;
    }
}

// One input row.  MODE 0 (row 2 s): pc = aP + x . w[6..8] (finished depthwise row s - 1 -> MFMAs into D when EMIT), aP <- x . w[0..2];
// MODE 1 (row 2 s + 1): aP += x . w[3..5].  src: LDS byte address of input pixel 2 p of this lane's tile, channel kq.
template <int CK, int MT, int PSV, int MODE, bool EMIT, bool TAPL>
__device__ __forceinline__ void ms2_row(const unsigned src, const unsigned aop, const float (&tap)[TAPL ? 1 : CK][9], float (&aP)[CK], df32x4 (&D)[MT], const float4* tapl) {
    float xs[2][3], av[2][MT];
    auto load_ks = [&](auto ksc, float (&x)[3], float (&aw)[MT]) {
        constexpr int ks = decltype(ksc)::value;
        const unsigned xa = src, aa = aop;
#pragma unroll
        for (int dx = 0; dx < 3; dx++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[dx]) : "v"(xa), "n"((dx * PSV + 4 * ks) * 4));
        if constexpr (MODE == 0 && EMIT) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(aw[mt]) : "v"(aa), "n"((ks * MT + mt) * 64 * 4));
        }
    };
    auto landed = [&](float (&x)[3], float (&aw)[MT]) {
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
        if constexpr (MODE == 0 && EMIT) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) asm volatile("" : "+v"(aw[mt]));
        }
    };
    auto kstep = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        if constexpr (ks + 1 < CK) load_ks(std::integral_constant<int, ks + 1>{}, xs[(ks + 1) & 1], av[(ks + 1) & 1]);
        float (&x)[3] = xs[ks & 1];
        float (&aw)[MT] = av[ks & 1];
        float w[9];
        if constexpr (TAPL) {
            const float4 t0 = tapl[ks * 12], t1 = tapl[ks * 12 + 1], t2 = tapl[ks * 12 + 2];
            w[0] = t0.x; w[1] = t0.y; w[2] = t0.z; w[3] = t0.w; w[4] = t1.x; w[5] = t1.y; w[6] = t1.z; w[7] = t1.w; w[8] = t2.x;
        } else {
#pragma unroll
            for (int t = 0; t < 9; t++) w[t] = tap[ks][t];
        }
        if constexpr (MODE == 0) {
            float pc = aP[ks], n = x[0] * w[0];
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                if (EMIT) pc = __builtin_fmaf(x[dx], w[6 + dx], pc);
                if (dx) n = __builtin_fmaf(x[dx], w[dx], n);
            }
            aP[ks] = n;
            asm volatile("" : "+v"(aP[ks]));
            if constexpr (EMIT) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) D[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt], pc, D[mt], 0, 0, 0);
            }
        } else {
            float c = aP[ks];
#pragma unroll
            for (int dx = 0; dx < 3; dx++) c = __builtin_fmaf(x[dx], w[3 + dx], c);
            aP[ks] = c;
            asm volatile("" : "+v"(aP[ks]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (ks + 1 < CK) landed(xs[(ks + 1) & 1], av[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
    };
    load_ks(std::integral_constant<int, 0>{}, xs[0], av[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    landed(xs[0], av[0]);
    __builtin_amdgcn_sched_barrier(0);
    dfor_each(kstep, std::make_integer_sequence<int, CK>{});
}

template <class K, bool RELU>
__global__ __launch_bounds__(K::NF * K::NWV * 64, 2) void ms2_kernel(Ms2Args a) {
    constexpr int CK = K::CK, MT = K::MT, CO = K::CO, NWV = K::NWV, NF = K::NF, WO = K::WO, WI = K::WI, C = K::C, PS = K::PS, QP = K::QP, IMG_F = K::IMG_F;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = wave / NWV, strip = wave - fr * NWV;
    const int kq = lane >> 4, p = lane & 15;
    float* wgc = lds;
    float* img = lds + K::TOTAL + fr * 4 * IMG_F;     // this frame's four row images
    const int groups = (a.B + NF - 1) / NF;
    const int band = blockIdx.x / groups, b = min((blockIdx.x - band * groups) * NF + fr, a.B - 1);   // (an odd batch's last group repeats the last frame)
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, a.Ho);
    const int Hi = 2 * a.Ho;
    const int xo0 = strip * 16;                        // this wave's output pixels [xo0, xo0 + 16), input pixels [2 xo0, 2 xo0 + 32)
    const int npx = min(32, WI - 2 * xo0);             // input pixels this wave fetches per row (16 for a half-used last tile)
    const int ngrp = npx / (4 * K::DPX);               // DMA groups of four instructions
    const float* in = a.in + (long)b * a.in_fs;

    const int goff = ((lane / QP) * C + 4 * min(lane % QP, CK - 1)) * 4;
    const unsigned lds_img = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)img);
    auto issue_row = [&](int r, int slot) {   // input row r (clamped: rows below the image are cleared after landing) -> image `slot`
        const char* src = reinterpret_cast<const char*>(in + ((long)min(r, Hi - 1) * WI + 2 * xo0) * C);
        unsigned dstb = lds_img + (unsigned)((slot * IMG_F + 2 * xo0 * PS) * 4);
        for (int gi = 0; gi < ngrp; gi++) {   // wave-uniform
            unsigned long long saved;
#define MI_MS2_DMA(k) "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 offset:" #k "*%6\n\t"
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, %4\n\t"
                         "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                         MI_MS2_DMA(1) MI_MS2_DMA(2) MI_MS2_DMA(3)
                         "s_mov_b64 exec, %0"
                         : "=&s"(saved)
                         : "s"(src), "v"(goff), "s"(dstb), "n"(K::ACTIVE == 64 ? 0xffffffffu : ((1u << (K::ACTIVE - 32)) - 1)), "n"(K::DPX * (PS - C) * 4), "n"(K::DPX * C * 4)
                         : "memory", "scc", "m0");
#undef MI_MS2_DMA
            src += 4 * K::DPX * C * 4;
            dstb += 4 * K::DPX * PS * 4;
        }
    };
    auto fix_row = [&](int r, int slot) {   // rows below the image: zeros
        if (r < Hi) return;  // wave-uniform
        float zz = 0.f;
        asm volatile("" : "+v"(zz));
        const float4 z = make_float4(zz, zz, zz, zz);
        float* part = img + slot * IMG_F + 2 * xo0 * PS;
        const int n4 = npx * QP;
        for (int k = lane; k < n4; k += 64) *reinterpret_cast<float4*>(part + 4 * k) = z;
    };
    // the zero column right of the last input pixel: never written afterwards
    if (strip == 0 && lane < QP) {
#pragma unroll
        for (int s = 0; s < 4; s++) *reinterpret_cast<float4*>(img + s * IMG_F + WI * PS + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // rows of the first step are on their way while the constants are fetched; those of the second go out behind them
    issue_row(2 * y0, 0);
    issue_row(2 * y0 + 1, 1);
    for (int i = threadIdx.x; i < K::TOTAL / 4; i += NF * NWV * 64) reinterpret_cast<float4*>(wgc)[i] = reinterpret_cast<const float4*>(a.consts)[i];
    if (y0 + 1 <= y1) { issue_row(2 * y0 + 2, 2); issue_row(2 * y0 + 3, 3); }
    // (the constants' loads are older than the second pair of rows: waiting for "at most that pair outstanding" covers them)
    ms2_wait_le(y0 + 1 <= y1 ? 8 * ngrp : 0);
    dwg_barrier();
    asm volatile("" ::: "memory");
    float tap[K::TAPL ? 1 : CK][9];
    if constexpr (!K::TAPL) {
        const float4* tp = reinterpret_cast<const float4*>(wgc + K::OFF_TAP) + kq * 3;
#pragma unroll
        for (int ks = 0; ks < CK; ks++) {
            const float4 t0 = tp[ks * 12], t1 = tp[ks * 12 + 1], t2 = tp[ks * 12 + 2];
            tap[ks][0] = t0.x; tap[ks][1] = t0.y; tap[ks][2] = t0.z; tap[ks][3] = t0.w;
            tap[ks][4] = t1.x; tap[ks][5] = t1.y; tap[ks][6] = t1.z; tap[ks][7] = t1.w;
            tap[ks][8] = t2.x;
        }
#pragma unroll
        for (int ks = 0; ks < CK; ks++)
#pragma unroll
            for (int t = 0; t < 9; t++) asm volatile("" : "+v"(tap[ks][t]));
    }
    // ---- per-lane addresses (a lane whose output pixel lies right of the frame works on the last real one and stores nothing)
    typedef __attribute__((address_space(3))) float lfloat;
    const int xo = min(xo0 + p, WO - 1);
    const bool live = xo0 + p < WO;
    const unsigned x_lds = (unsigned)(uintptr_t)(lfloat*)(img + 2 * xo * PS + kq);           // B layout: input pixel 2 xo, channel kq
    const unsigned a_lds = (unsigned)(uintptr_t)(lfloat*)(wgc + K::OFF_A + lane);
    const float* sme = img + 2 * xo * PS + 4 * kq;                                            // D layout: input pixel 2 xo, channels 4 kq .. of a 16-channel tile
    const unsigned ooff = (unsigned)(xo * CO + 4 * kq) * 4u;
    const float4* tapl = reinterpret_cast<const float4*>(wgc + K::OFF_TAP) + kq * 3;

    float aP[CK];
#pragma unroll
    for (int ks = 0; ks < CK; ks++) aP[ks] = 0.f;
    df32x4 D[MT];
    // D <- bias + max-pool skip of the input rows in images s0 / s0 + 1 (channels >= C: the zero channel-pad)
    auto init_D = [&](int s0) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const float4 bs = *reinterpret_cast<const float4*>(wgc + K::OFF_BIAS + 16 * mt + 4 * kq);
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
            if (K::SKIP && 16 * mt < C) {
                const float* r0 = sme + s0 * IMG_F + 16 * mt;
                const float4 q00 = *reinterpret_cast<const float4*>(r0), q01 = *reinterpret_cast<const float4*>(r0 + PS);
                const float4 q10 = *reinterpret_cast<const float4*>(r0 + IMG_F), q11 = *reinterpret_cast<const float4*>(r0 + IMG_F + PS);
                m = make_float4(fmaxf(fmaxf(q00.x, q01.x), fmaxf(q10.x, q11.x)), fmaxf(fmaxf(q00.y, q01.y), fmaxf(q10.y, q11.y)),
                                fmaxf(fmaxf(q00.z, q01.z), fmaxf(q10.z, q11.z)), fmaxf(fmaxf(q00.w, q01.w), fmaxf(q10.w, q11.w)));
            }
            D[mt] = df32x4{bs.x + m.x, bs.y + m.y, bs.z + m.z, bs.w + m.w};
        }
    };
    typedef __attribute__((address_space(1))) char gchar;
    typedef __attribute__((address_space(1))) df32x4 gf32x4;
    auto epilogue = [&](int y) {
        gchar* dst = (gchar*)(a.out + (long)b * a.out_fs + (long)y * WO * CO);
        asm volatile("" : "+s"(dst));
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            float4 sl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!RELU) sl = *reinterpret_cast<const float4*>(wgc + K::OFF_SLOPE + 16 * mt + 4 * kq);
            df32x4 v = D[mt];
            if (RELU) {
                v = df32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
            } else {
                v = df32x4{fminf(fmaxf(v.x, 0.f) + sl.x * fminf(v.x, 0.f), a.hi), fminf(fmaxf(v.y, 0.f) + sl.y * fminf(v.y, 0.f), a.hi),
                           fminf(fmaxf(v.z, 0.f) + sl.z * fminf(v.z, 0.f), a.hi), fminf(fmaxf(v.w, 0.f) + sl.w * fminf(v.w, 0.f), a.hi)};
            }
            if (live && 16 * mt + 4 * kq < CO) *(gf32x4*)(dst + ooff + (unsigned)(16 * mt * 4)) = v;
        }
    };
    // vmcnt bookkeeping (operations retire in issue order): a step's operations are the stores of output row s - 1, then the DMA of the two
    // rows of step s + 2.  "The rows of step s have landed" = at most the DMA issued at the end of step s - 1 is outstanding.
    auto step = [&](auto emit, int s) {
        constexpr bool EMIT = decltype(emit)::value;
        const int s0 = 2 * ((s - y0) & 1);   // band-relative: the prologue put step y0's rows into images 0 / 1 (round 5: a band that starts on an odd
                                              // output row — odd band sizes, first met at 70 ROIs — read the wrong pair)
        const bool dm = s == y0 || s + 1 <= y1;   // step y0: the prologue's second pair of rows is behind the first
        ms2_wait_le(dm ? 8 * ngrp : 0);           // (step s - 1's stores are older than its DMA: they are waited for as well)
        fix_row(2 * s, s0);
        fix_row(2 * s + 1, s0 + 1);
        dwg_barrier();   // both rows are complete (every wave's part has landed and is fixed)
        __builtin_amdgcn_sched_barrier(0);
        ms2_row<CK, MT, PS, 0, EMIT, K::TAPL>(x_lds + (unsigned)(s0 * IMG_F * 4), a_lds, tap, aP, D, tapl);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (EMIT) epilogue(s - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (s < y1) {   // uniform: output row s exists
            ms2_row<CK, MT, PS, 1, false, K::TAPL>(x_lds + (unsigned)((s0 + 1) * IMG_F * 4), a_lds, tap, aP, D, tapl);
            __builtin_amdgcn_sched_barrier(0);
            init_D(s0);
        }
        dwg_barrier();   // everybody is done with the two rows (the neighbours' reads of this wave's pixels included)
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 <= y1) { issue_row(2 * (s + 2), s0); issue_row(2 * (s + 2) + 1, s0 + 1); }
    };
    step(std::false_type{}, y0);
    for (int s = y0 + 1; s <= y1; s++) step(std::true_type{}, s);
}

using MS2a = MS2<8, 4, 64, 2, 24, 2, true, false>;    // face mesh: 48 x 48 x 32 -> 24 x 24 x 64, max-pool skip, two frames per workgroup
using MS2b = MS2<8, 1, 12, 3, 48, 1, false, false>;   // full_range: 96 x 96 x 32 -> 48 x 48 x 12
using MS2c = MS2<16, 2, 24, 2, 24, 1, false, true>;   // full_range: 48 x 48 x 64 -> 24 x 24 x 24, taps in LDS
using MS2d = MS2<4, 2, 32, 3, 48, 2, true, false>;    // face mesh (round 5): 96 x 96 x 16 -> 48 x 48 x 32, max-pool skip, two frames per workgroup

template <class K>
void ms2_pack(const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst) {
    constexpr int C = K::C, CO = K::CO;
    std::fill(dst, dst + K::TOTAL, 0.f);
    for (int ks = 0; ks < K::CK; ks++)
        for (int mt = 0; mt < K::MT; mt++)
            for (int l = 0; l < 64; l++) {
                const int o = 16 * mt + (l & 15);
                if (o < CO) dst[K::OFF_A + (ks * K::MT + mt) * 64 + l] = w_pw[(size_t)o * C + 4 * ks + (l >> 4)];
            }
    for (int ks = 0; ks < K::CK; ks++)
        for (int kq = 0; kq < 4; kq++)
            for (int t = 0; t < 9; t++) dst[K::OFF_TAP + (ks * 4 + kq) * 12 + t] = w_dw[t * C + 4 * ks + kq];
    for (int c = 0; c < CO; c++) {
        double acc = bias ? bias[c] : 0.0;
        if (b_dw)
            for (int k = 0; k < C; k++) acc += (double)w_pw[(size_t)c * C + k] * b_dw[k];
        dst[K::OFF_BIAS + c] = (float)acc;
        dst[K::OFF_SLOPE + c] = act == ACT_PRELU ? alpha[c] : (act == ACT_NONE ? 1.f : 0.f);
    }
}

template <class K>
int ms2_launch(const BlockArgs& a, hipStream_t s) {
    Ms2Args ma;
    ma.in = a.in; ma.out = a.out; ma.consts = a.w_mwalk; ma.in_fs = a.in_fs; ma.out_fs = a.out_fs;
    ma.B = a.B; ma.Ho = a.Ho;
    ma.hi = a.ep.act == ACT_RELU6 ? 6.f : INFINITY;
    static const int forced = getenv("MI_MS2_BAND") ? atoi(getenv("MI_MS2_BAND")) : 0;  // tuning aid (output rows per band)
    const int groups = (a.B + K::NF - 1) / K::NF;
    const int per_cu = std::max(1, std::min(8 / (K::NF * K::NWV), (int)((160 * 1024) / (K::LDS_F * 4))));
    long bands = std::max<long>(1, ((long)per_cu * device_cu_count() + groups / 2) / std::max(1, groups));
    int rows = (int)((a.Ho + bands - 1) / bands);
    rows = std::max(rows, std::min(a.Ho, 6));
    if (forced > 0) rows = std::min(forced, a.Ho);
    ma.band_rows = rows;
    ma.bands = (a.Ho + rows - 1) / rows;
    const dim3 grid((unsigned)((long)groups * ma.bands));
    const size_t lds_bytes = (size_t)K::LDS_F * 4;
    if (a.ep.act == ACT_RELU) {
        auto kern = ms2_kernel<K, true>;
        if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
        return (int)launch_kernel(kern, grid, dim3(K::NF * K::NWV * 64), lds_bytes, s, ma);
    }
    auto kern = ms2_kernel<K, false>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    return (int)launch_kernel(kern, grid, dim3(K::NF * K::NWV * 64), lds_bytes, s, ma);
}

// 0: none; 1: 48 wide 32 -> 64 with the max-pool skip; 2: 96 wide 32 -> 12, no skip; 3: 48 wide 64 -> 24, no skip; 4: 96 wide 16 -> 32 with the skip  (W = INPUT width)
int ms2_shape(int W, int C, int Co, bool skip) {
    static const bool off = getenv("MI_NO_MS2") != nullptr;  // tuning aid: the block kernel instead
    if (off) return 0;
    if (W == 48 && C == 32 && Co == 64 && skip) return 1;
    if (W == 96 && C == 32 && Co == 12 && !skip) return 2;
    if (W == 48 && C == 64 && Co == 24 && !skip) return 3;
    if (W == 96 && C == 16 && Co == 32 && skip) return 4;
    return 0;
}

}  // namespace

bool ms2_shape_ok(int W, int C, int Co, bool skip) { return ms2_shape(W, C, Co, skip) != 0; }

int ms2_consts_floats(int W, int C, int Co, bool skip) {
    switch (ms2_shape(W, C, Co, skip)) {
        case 1: return MS2a::TOTAL;
        case 2: return MS2b::TOTAL;
        case 3: return MS2c::TOTAL;
        case 4: return MS2d::TOTAL;
    }
    return 0;
}

// w_dw [3][3][C], b_dw [C] or null, w_pw [Co][C], bias [Co] or null, alpha [Co] or null
void ms2_pack_consts(int W, int C, int Co, bool skip, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst) {
    switch (ms2_shape(W, C, Co, skip)) {
        case 1: ms2_pack<MS2a>(w_dw, b_dw, w_pw, bias, alpha, act, dst); break;
        case 2: ms2_pack<MS2b>(w_dw, b_dw, w_pw, bias, alpha, act, dst); break;
        case 3: ms2_pack<MS2c>(w_dw, b_dw, w_pw, bias, alpha, act, dst); break;
        case 4: ms2_pack<MS2d>(w_dw, b_dw, w_pw, bias, alpha, act, dst); break;
    }
}

// stride-2 depthwise 3x3 (TFLite SAME on an even-sized frame: no pad before) + pointwise, skip = 2x2 max-pool of the block's own input or none
bool ms2_kernel_supports(const BlockArgs& a) {
    if (!a.w_mwalk || !a.has_dw || a.sh != 2 || a.sw != 2 || a.pt != 0 || a.pl != 0) return false;
    if ((a.H & 1) || (a.W & 1) || a.Ho * 2 != a.H || a.Wo * 2 != a.W || a.Ho < 2) return false;
    const bool skip = a.ep.res_mode != RES_NONE;
    if (skip) {
        if (a.ep.res_after || a.ep.res_mode != RES_MAXPOOL || a.ep.res != a.in || a.ep.res_fs != a.in_fs || a.ep.res_C != a.C) return false;
    }
    if (!ms2_shape(a.W, a.C, a.Co, skip)) return false;
    static const int min_b = getenv("MI_MS2_MIN_B") ? atoi(getenv("MI_MS2_MIN_B")) : 32;
    if (a.B < min_b) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return aligned16(a.in) && aligned16(a.out) && aligned16(a.w_mwalk) && !(a.in_fs & 3) && !(a.out_fs & 3);
}

const char* ms2_kernel_label(const BlockArgs& a, char* buf, size_t cap) {
    snprintf(buf, cap, "ms2_kernel<%d,%d,%d>", a.C / 4, (a.Co + 15) / 16, (a.Wo + 15) / 16);
    return buf;
}

int launch_ms2(const BlockArgs& a, void* stream) {
    if (!ms2_kernel_supports(a)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    switch (ms2_shape(a.W, a.C, a.Co, a.ep.res_mode != RES_NONE)) {
        case 1: return ms2_launch<MS2a>(a, s);
        case 2: return ms2_launch<MS2b>(a, s);
        case 3: return ms2_launch<MS2c>(a, s);
        case 4: return ms2_launch<MS2d>(a, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace mi
