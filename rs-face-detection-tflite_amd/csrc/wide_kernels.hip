// wide_kernels.hip — register-resident fused BlazeBlocks for 32-pixel-wide layers with more channels than one wave's SGPRs
// can feed (BackCamera's seven 32x32x48 blocks):
//
//   out = act( PW1x1( DW3x3(in) + b_dw ) + b_pw + in )
//
// Same operator chain as strip_kernels.hip (DEPTHWISE_CONV_2D -> CONV_2D 1x1 -> ADD -> RELU behind `interpreter.invoke()`,
// /root/reference/src/face_detection_lite/face_detection.rs:235; graph: SURVEY.md Appendix A.1, "7 x Block(48->48, s1)"), and the
// same mapping idea — lane = pixel, vertical reuse in registers, every weight an SGPR-pair operand of v_pk_fma_f32 streamed
// through the scalar cache — stretched in two directions:
//   * a 32-pixel row fills half a wave, so a 64-lane strip is TWO FRAMES side by side (lanes 0-31 frame 2p, lanes 32-63
//     frame 2p+1); nothing else changes for a lane, whose left / right neighbours are still the adjacent pixels of its row
//     image (each frame's row has its own zero border columns in LDS).
//   * a stage (one input-channel pair) needs 18 depthwise taps + 2 pointwise columns in SGPRs.  With 48 output channels the
//     columns alone are 96 floats: over the 102-SGPR budget.  So the OUTPUT channels are split over the NS = 2 waves of a
//     workgroup: each wave runs the depthwise stage for all 48 input channels (9 of a stage's 33 packed FMAs, recomputed by
//     the partner: +16 % arithmetic) and the pointwise conv, bias, skip and activation for its own 24 output channels
//     (stage record = 18 + 2 x 24 floats, exactly the 24-channel strip kernel's).  The two waves share the row images: each
//     brings in one frame's half by LDS-DMA and one s_barrier per row publishes both halves.
//   * wave h walks the input channels starting at its own output half (quads h*CQO .. then the rest), so the centre pixels
//     it meets first are its skip connection; only the order of the pointwise sum differs between the halves.
// Row image in LDS: [2 frames][34 pixels][C + 4 floats]: the pixel stride of 52 floats makes the lanes' ds_read_b128
// conflict-free (48 would be 4-way: 48 l mod 64 takes 4 values).  LDS-DMA destinations are lane-linear and sources per lane,
// so one instruction brings in 4 pixels with 52 lanes (the 13th float4 slot of a pixel gets a repeated quad): every
// instruction of a row has the same per-lane source offset, the pixel group moves in the scalar base.
// Exact f32; results match the block kernel to reassociation of the sums and the folded depthwise bias.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"
#include "launch.hpp"

namespace mi {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float __attribute__((address_space(4))) cfloat;

struct WideArgs {
    const float* in;
    float* out;
    const float* consts;   // wide_pack_consts(): NS blobs of WK::TOTAL floats, one per output-channel share
    long in_fs, out_fs;
    int B, H;              // W == 32
    int pairs, bands, band_rows;
    int has_res;
    float hi;              // upper clamp of the activation (6 for ReLU6, +inf otherwise)
};

template <int CQI, int NS>
struct WK {
    static constexpr int CQO = CQI / NS;       // output channel quads per wave
    static constexpr int C = 4 * CQI, CO = 4 * CQO;
    static constexpr int PS = C + 4;           // pixel stride of the row image (floats)
    static constexpr int QP = CQI + 1;         // float4 slots per pixel
    static constexpr int IMG_F = 68 * PS;      // one row image: two frames of 34 pixels
    static constexpr int DPX = 64 / QP;        // pixels one LDS-DMA instruction brings in (DPX * QP of the 64 lanes active)
    static constexpr int NLD = 32 / DPX;       // LDS-DMA instructions per frame row
    static_assert(32 % DPX == 0, "a frame row is a whole number of DMA instructions");
    static constexpr int NST = 2 * CQI;        // stages (input channel pairs)
    static constexpr int ST_F = (32 + 2 * CO + 15) / 16 * 16;
    static constexpr int OFF_BIAS = NST * ST_F, OFF_SLOPE = OFF_BIAS + 32, TOTAL = OFF_SLOPE + 32;
    static constexpr int OBUF_F = 64 * CO;     // transposition buffer of one wave
};

__device__ __forceinline__ float4 wld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void wst4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void wwave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ v2f wpkfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
template <int N>
__device__ __forceinline__ void wwait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70); }

// One input row r of the lane's pixel: see strip_row (strip_kernels.hip) — the same stage structure (wait | load the 2nd
// channel's pointwise column | depthwise + 1st channel's pointwise | wait | load the next stage, next quad's pixels | 2nd
// channel's pointwise), here over 2*CQI input-channel pairs into CQO output quads.  pA / pB: the lane's left neighbour in the
// row image, offset to the first quad of the wave's own half / of the other quads (walk order: own half first).
// oacc arrives holding skip + bias of the output row this call finishes (wide_init, called behind the previous row: the skip is
// that row's centre pixels, which this call's predecessor collected in xcur — one set of centre pixels live instead of two).
template <int CQI, int CQO, bool EMIT>
__device__ __forceinline__ void wide_row(const float* pA, const float* pB, const cfloat* cst, v2f (&aPN)[CQI][2], v2f (&aC)[CQI][2],
                                         float4 (&xcur)[CQO], v2f (&oacc)[2 * CQO]) {
    constexpr int C = 4 * CQI, CO = 4 * CQO, PS = C + 4;
    constexpr int ST_F = (32 + 2 * CO + 15) / 16 * 16;
    float wd[18], wp0[CO], wp1[CO];
    float4 xbuf[2][3];
#ifdef MI_ABL_NOSMEM  // timing ablation (development harness only): weights from one scalar register, no scalar loads
    float abl_w = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, aC[0][0].x)));
    auto load_first = [&](int st) {
        asm volatile("" : "+s"(abl_w));
#pragma unroll
        for (int i = 0; i < 18; i++) wd[i] = abl_w;
#pragma unroll
        for (int i = 0; i < CO; i++) wp0[i] = abl_w;
    };
    auto load_second = [&](int st) {
        asm volatile("" : "+s"(abl_w));
#pragma unroll
        for (int i = 0; i < CO; i++) wp1[i] = abl_w;
    };
#else
    auto load_first = [&](int st) {
        const cfloat* p = cst;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int i = 0; i < 18; i++) wd[i] = p[st * ST_F + i];
        if (EMIT) {
#pragma unroll
            for (int i = 0; i < CO; i++) wp0[i] = p[st * ST_F + 32 + i];
        }
    };
    auto load_second = [&](int st) {
        const cfloat* p = cst;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int i = 0; i < CO; i++) wp1[i] = p[st * ST_F + 32 + CO + i];
    };
#endif
    auto load_x = [&](int q, float4 (&x)[3]) {
        const float* base = q < CQO ? pA : pB;
#pragma unroll
        for (int kx = 0; kx < 3; kx++) x[kx] = wld4(base + kx * PS + 4 * q);
    };
    load_first(0);
    load_x(0, xbuf[0]);
#pragma unroll
    for (int st = 0; st < 2 * CQI; st++) {
        const int q = st >> 1, h = st & 1;
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        if (EMIT) load_second(st);
        __builtin_amdgcn_sched_barrier(0);
        const float4 (&x)[3] = xbuf[q & 1];
        if (h == 0 && q < CQO) xcur[q] = x[1];
        v2f t[3];
#pragma unroll
        for (int kx = 0; kx < 3; kx++) t[kx] = h == 0 ? v2f{x[kx].x, x[kx].y} : v2f{x[kx].z, x[kx].w};
        auto wv = [&](int tap) { return v2f{wd[2 * tap], wd[2 * tap + 1]}; };
        v2f n = t[0] * wv(0), c = aC[q][h], pch = aPN[q][h];
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            if (EMIT) pch = wpkfma(t[kx], wv(6 + kx), pch);
            if (kx) n = wpkfma(t[kx], wv(kx), n);
            c = wpkfma(t[kx], wv(3 + kx), c);
        }
        aC[q][h] = c;
        aPN[q][h] = n;
        asm volatile("" : "+v"(aC[q][h]), "+v"(aPN[q][h]));
        if constexpr (EMIT) {
#pragma unroll
            for (int o = 0; o < 2 * CQO; o++) oacc[o] = wpkfma(v2f{pch.x, pch.x}, v2f{wp0[2 * o], wp0[2 * o + 1]}, oacc[o]);
            asm volatile("" : "+v"(pch));
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (st + 1 < 2 * CQI) {
            load_first(st + 1);
            if (h == 1) load_x(q + 1, xbuf[(q + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (EMIT) {
#pragma unroll
            for (int o = 0; o < 2 * CQO; o++) {
                oacc[o] = wpkfma(v2f{pch.y, pch.y}, v2f{wp1[2 * o], wp1[2 * o + 1]}, oacc[o]);
                asm volatile("" : "+v"(oacc[o]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// accumulators of the NEXT output row: skip (the centre pixels just collected) + bias (= b_pw + W b_dw), in that order
template <int CQO>
__device__ __forceinline__ void wide_init(const cfloat* bias, bool has_res, const float4 (&x)[CQO], v2f (&oacc)[2 * CQO]) {
    const cfloat* bp = bias;
    asm volatile("" : "+s"(bp));
    if (has_res) {  // wave-uniform branch
#pragma unroll
        for (int q = 0; q < CQO; q++) {
            oacc[2 * q] = v2f{x[q].x, x[q].y} + v2f{bp[4 * q], bp[4 * q + 1]};
            oacc[2 * q + 1] = v2f{x[q].z, x[q].w} + v2f{bp[4 * q + 2], bp[4 * q + 3]};
        }
    } else {
#pragma unroll
        for (int q = 0; q < CQO; q++) {
            oacc[2 * q] = v2f{bp[4 * q], bp[4 * q + 1]};
            oacc[2 * q + 1] = v2f{bp[4 * q + 2], bp[4 * q + 3]};
        }
    }
#pragma unroll
    for (int o = 0; o < 2 * CQO; o++) asm volatile("" : "+v"(oacc[o]));  // keeps the two arms from being merged into selects
}

template <int CQO, bool RELU>
__device__ __forceinline__ void wide_act(const v2f (&oacc)[2 * CQO], const cfloat* slopes, float hi, float4 (&o)[CQO]) {
#pragma unroll
    for (int q = 0; q < CQO; q++) {
        o[q] = make_float4(oacc[2 * q].x, oacc[2 * q].y, oacc[2 * q + 1].x, oacc[2 * q + 1].y);
        if (RELU) {
            o[q].x = fmaxf(o[q].x, 0.f); o[q].y = fmaxf(o[q].y, 0.f); o[q].z = fmaxf(o[q].z, 0.f); o[q].w = fmaxf(o[q].w, 0.f);
        } else {
            const cfloat* sq = slopes + 4 * q;
            o[q].x = fminf(fmaxf(o[q].x, 0.f) + sq[0] * fminf(o[q].x, 0.f), hi);
            o[q].y = fminf(fmaxf(o[q].y, 0.f) + sq[1] * fminf(o[q].y, 0.f), hi);
            o[q].z = fminf(fmaxf(o[q].z, 0.f) + sq[2] * fminf(o[q].z, 0.f), hi);
            o[q].w = fminf(fmaxf(o[q].w, 0.f) + sq[3] * fminf(o[q].w, 0.f), hi);
        }
    }
}

// One workgroup = NS waves = one band of rows of one frame pair.  NBUF row images: row r is read while rows r+1 .. r+NBUF-1 are
// in flight (a buffer can only be refilled once BOTH waves have left the row that was in it, i.e. behind the next barrier).
template <int CQI, int NS, bool RELU, int NBUF>
__global__ __launch_bounds__(64 * NS, NS == 2 ? 2 : 1) void wide_kernel(WideArgs a) {
    using K = WK<CQI, NS>;
    constexpr int C = K::C, CQO = K::CQO, CO = K::CO, PS = K::PS, QP = K::QP, NLD = K::NLD, IMG_F = K::IMG_F;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int h = NS == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // output-channel share = wave
    const int fp = blockIdx.x % a.pairs, band = blockIdx.x / a.pairs;  // bands of a pair are `pairs` ids apart: one XCD when pairs % 8 == 0
    const int f = lane >> 5, px = lane & 31;
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, a.H);
    if (y0 >= y1) return;  // whole workgroup
    float* img = lds;                                   // [NBUF][IMG_F]
    float* obuf = lds + NBUF * IMG_F + h * K::OBUF_F;   // wave-private
#ifdef MI_ABL_SAMEBLOB  // timing ablation (development harness only): both waves stream the same constants
    const cfloat* cst = (const cfloat*)a.consts;
#else
    const cfloat* cst = (const cfloat*)a.consts + h * K::TOTAL;
#endif

    // ---- DMA of one frame row into a row image: instruction k brings in pixels [DPX k, DPX k + DPX): lane -> (pixel lane / QP,
    // quad min(lane % QP, CQI - 1)), lanes >= DPX * QP idle.  Wave h brings in frame h of the pair (NS == 1: both).
    const int goff = ((lane / QP) * C + 4 * min(lane % QP, CQI - 1)) * 4;  // bytes from the pixel group's start
    const unsigned lds_img = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)img);
    auto issue_frame_row = [&](int r, int bi, int fr) {
        const int b = min(2 * fp + fr, a.B - 1);
        const char* src = reinterpret_cast<const char*>(a.in + (long)b * a.in_fs + (long)min(max(r, 0), a.H - 1) * 32 * C);
        const unsigned dstb = lds_img + (unsigned)((bi * IMG_F + (fr * 34 + 1) * PS) * 4);
        constexpr int ACTIVE = K::DPX * QP;
        static_assert(ACTIVE > 32 && ACTIVE < 64, "exec mask written as two 32-bit halves (SALU literals are 32 bits)");
        static_assert(NLD == 8 && (NLD / 2 - 1) * K::DPX * C * 4 < 4096, "two source bases, four immediate offsets each");
        // ONE asm statement: exec is narrowed to the active lanes around the eight instructions, and nothing the compiler
        // schedules may land in between (a VALU instruction there would skip lanes 52-63)
        const char* src1 = src + (NLD / 2) * K::DPX * C * 4;
        unsigned long long saved;
        // the instruction's immediate offset moves the LDS destination as well as the source: M0 only makes up the difference
        // between the image's pixel-group stride and the source's (DPX * (PS - C) floats)
#define MI_WIDE_DMA(base, k) "s_add_u32 m0, m0, %7\n\tglobal_load_lds_dwordx4 %3, " base " offset:" #k "*%8\n\t"
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, %6\n\t"
                     "s_mov_b32 m0, %4\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     MI_WIDE_DMA("%1", 1) MI_WIDE_DMA("%1", 2) MI_WIDE_DMA("%1", 3)
                     "s_add_u32 m0, m0, %5\n\tglobal_load_lds_dwordx4 %3, %2\n\t"
                     MI_WIDE_DMA("%2", 1) MI_WIDE_DMA("%2", 2) MI_WIDE_DMA("%2", 3)
                     "s_mov_b64 exec, %0"
                     : "=&s"(saved)
                     : "s"(src), "s"(src1), "v"(goff), "s"(dstb), "n"((NLD / 2) * K::DPX * PS * 4 - (NLD / 2 - 1) * K::DPX * (PS - C) * 4),
                       "n"((1u << (ACTIVE - 32)) - 1), "n"(K::DPX * (PS - C) * 4), "n"(K::DPX * C * 4)
                     : "memory", "scc");
#undef MI_WIDE_DMA
    };
    constexpr int NDMA = NS == 1 ? 2 * NLD : NLD;  // DMA instructions one wave issues per row
    auto issue_row = [&](int r, int bi) {
        if (NS == 1) { issue_frame_row(r, bi, 0); issue_frame_row(r, bi, 1); }
        else issue_frame_row(r, bi, h);
    };
    // rows outside the image are zero padding: the landed (clamped) row is cleared before it is published
    auto fix_row = [&](int r, int bi) {
        if (r >= 0 && r < a.H) return;  // wave-uniform
        float zz = 0.f;
        asm volatile("" : "+v"(zz));    // made here: a zero vector kept live across the row loop costs four VGPRs (or a spill)
        const float4 z = make_float4(zz, zz, zz, zz);
#pragma unroll
        for (int fr = 0; fr < 2; fr++) {
            if (NS == 2 && fr != h) continue;
            float* part = img + bi * IMG_F + (fr * 34 + 1) * PS;
            constexpr int N4 = 32 * QP;
#pragma unroll
            for (int k = 0; k < (N4 + 63) / 64; k++)
                if (64 * (k + 1) <= N4 || lane < N4 - 64 * k) wst4(part + 4 * (lane + 64 * k), z);
        }
    };
    // the border pixel columns of every image (left of x = 0, right of x = 31, per frame) are never written by the DMA: cleared once
    {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = threadIdx.x; i < NBUF * 4 * QP; i += 64 * NS) {
            const int bi = i / (4 * QP), e = i - bi * 4 * QP, col = e / QP, qd = e - col * QP;
            const int pxi = (col >> 1) * 34 + (col & 1) * 33;
            wst4(img + bi * IMG_F + pxi * PS + 4 * qd, z);
        }
    }

    const float* me = img + (f * 34 + px) * PS;     // the lane's left neighbour in row image 0
    const int offA = 4 * (h * CQO), offB = -4 * (h * CQO);
    // output: float4 j = lane + 64 k of the wave's [64 pixels][CQO quads] transposition buffer -> (frame, column, quad); the 32
    // pixels of a frame are CQO / 2 whole instructions, so the second frame's offsets are the first's + one frame stride
    static_assert(CQO % 2 == 0, "half of the store instructions per frame");
    unsigned ooff[CQO / 2];  // bytes
#pragma unroll
    for (int k = 0; k < CQO / 2; k++) {
        const int j = lane + 64 * k, pl = j / CQO, qd = j - pl * CQO;
        ooff[k] = (unsigned)(pl * C + h * CO + 4 * qd) * 4u;
    }
    const bool pair_full = 2 * fp + 1 < a.B;  // wave-uniform
    float* out = a.out + (long)(2 * fp) * a.out_fs;

    v2f oacc[2 * CQO];
    v2f acc0[CQI][2], acc1[CQI][2];
    float4 xc[CQO];
#pragma unroll
    for (int q = 0; q < CQI; q++) acc0[q][0] = acc0[q][1] = acc1[q][0] = acc1[q][1] = v2f{0.f, 0.f};

    auto epilogue = [&](int y) {
        float4 o[CQO];
        wide_act<CQO, RELU>(oacc, cst + K::OFF_SLOPE, a.hi, o);
#pragma unroll
        for (int q = 0; q < CQO; q++) wst4(obuf + lane * CO + 4 * q, o[q]);
        wwave_sync();
        // scalar row base + 32-bit lane offset (opaque, or the loop optimiser carries a 64-bit pointer per store instruction in VGPRs)
        typedef __attribute__((address_space(1))) char gchar;      // global address space kept explicit: an opaque generic pointer
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(1))) f32x4 gfloat4;   // would make these FLAT stores (they wait on both counters)
        auto ldo = [&](int k) { const float4 v = wld4(obuf + 4 * lane + 256 * k); return f32x4{v.x, v.y, v.z, v.w}; };
        gchar* dst = (gchar*)(out + (long)y * 32 * C);
        asm volatile("" : "+s"(dst));
#pragma unroll
        for (int k = 0; k < CQO / 2; k++) *(gfloat4*)(dst + ooff[k]) = ldo(k);
        if (pair_full) {
            gchar* dst1 = dst + a.out_fs * 4;
            asm volatile("" : "+s"(dst1));
#pragma unroll
            for (int k = 0; k < CQO / 2; k++) *(gfloat4*)(dst1 + ooff[k]) = ldo(k + CQO / 2);
        }
        wwave_sync();
    };
    // vmcnt bookkeeping: vector-memory operations retire in issue order, so "row r has landed" = at most the operations this
    // wave issued after that row's DMA are still outstanding.  Step t (row y0 - 1 + t) issues the DMA of row y0 - 1 + t + NBUF - 1
    // when the band has it (t <= t_dma), then, from t = 2 on, the stores of an output row.
    const int n_st = pair_full ? CQO : CQO / 2;       // store instructions per output row
    const int t_dma = y1 - y0 + 2 - NBUF;             // last step that issues a DMA
    auto wait_younger = [&](int younger) {            // the largest counted wait that is still safe
        if (younger >= 2 * CQO + NDMA) wwait_vm<2 * CQO + NDMA>();
        else if (younger >= CQO + NDMA) wwait_vm<CQO + NDMA>();
        else if (younger >= NDMA) wwait_vm<NDMA>();
        else if (younger >= CQO) wwait_vm<CQO>();
        else if (younger >= CQO / 2) wwait_vm<CQO / 2>();
        else wwait_vm<0>();
    };
    auto ops_of_step = [&](int t, bool with_dma) { return (with_dma && t <= t_dma ? NDMA : 0) + (t >= 2 ? n_st : 0); };

    auto step = [&](auto emit, int r, v2f (&aPN)[CQI][2], v2f (&aC)[CQI][2]) {
        const int t = r - (y0 - 1);          // step number
        const int bi = t % NBUF;
        // operations younger than DMA(row r): that DMA was issued in step t - (NBUF - 1) ahead of that step's stores (or, for the
        // first NBUF - 1 rows, in the prologue ahead of the later prologue rows); every later step's DMA and stores follow
        int younger = 0;
        if (t >= NBUF - 1) {
            younger = ops_of_step(t - (NBUF - 1), false);
#pragma unroll
            for (int i = 1; i < NBUF - 1; i++) younger += ops_of_step(t - i, true);
        } else {
            younger = (NBUF - 2 - t) * NDMA;
#pragma unroll
            for (int i = 0; i < NBUF - 1; i++) younger += i < t ? ops_of_step(i, true) : 0;
        }
        wait_younger(younger);
        fix_row(r, bi);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the clears above, and every read of the previous row
#ifndef MI_ABL_NOBAR
        if (NS > 1) __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        if (t <= t_dma) issue_row(r + NBUF - 1, (t + NBUF - 1) % NBUF);
        const float* m = me + bi * IMG_F;
        wide_row<CQI, CQO, decltype(emit)::value>(m + offA, m + offB, cst, aPN, aC, xc, oacc);
        if constexpr (decltype(emit)::value) epilogue(r - 1);
        wide_init<CQO>(cst + K::OFF_BIAS, a.has_res != 0, xc, oacc);  // output row r starts from its skip (row r's centre pixels) + bias
    };
    // prologue: rows y0-1 .. y0+NBUF-3 (NBUF-1 rows) in flight before the first step
#pragma unroll
    for (int i = 0; i < NBUF - 1; i++)
        if (y0 - 1 + i <= y1) issue_row(y0 - 1 + i, i);
    step(std::false_type{}, y0 - 1, acc0, acc1);
    step(std::false_type{}, y0, acc1, acc0);
    for (int r = y0 + 1; r <= y1; r += 2) {
        step(std::true_type{}, r, acc0, acc1);
        if (r + 1 > y1) break;
        step(std::true_type{}, r + 1, acc1, acc0);
    }
}

template <int CQI, int NS, bool RELU, int NBUF>
int launch_wide_inst(const WideArgs& wa, hipStream_t s) {
    using K = WK<CQI, NS>;
    auto kern = wide_kernel<CQI, NS, RELU, NBUF>;
    const size_t lds_bytes = (size_t)(NBUF * K::IMG_F + NS * K::OBUF_F) * 4;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    return (int)launch_kernel(kern, dim3((unsigned)(wa.pairs * wa.bands)), dim3(64 * NS), lds_bytes, s, wa);
}

}  // namespace

// Shapes the wide kernel takes: stride-1 depthwise 3x3 (SAME) + pointwise with C = Co = 48 on 32-pixel-wide frames, skip = the
// block's own input (or none), constants packed by wide_pack_consts().
bool wide_shape_ok(int C, int Co) { return C == Co && C == 48 && getenv("MI_NO_WIDE") == nullptr; }

bool wide_kernel_supports(const BlockArgs& a) {
    if (!a.w_strip || !a.has_dw || a.sh != 1 || a.sw != 1 || a.pt != 1 || a.pl != 1) return false;
    if (!wide_shape_ok(a.C, a.Co) || a.W != 32 || a.H != a.Ho || a.W != a.Wo || a.H < 2) return false;
    if (a.ep.res_mode != RES_NONE) {
        if (a.ep.res_after) return false;
        if (a.ep.res_mode != RES_DIRECT || a.ep.res != a.in || a.ep.res_fs != a.in_fs || a.ep.res_C != a.C) return false;
    }
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!aligned16(a.in) || !aligned16(a.out) || (a.in_fs & 3) || (a.out_fs & 3)) return false;
    return a.out_fs < (1L << 28);
}

int wide_consts_floats(int C) { return C == 48 ? 2 * WK<12, 2>::TOTAL : 0; }

// w_dw [3][3][C], b_dw [C] or null, w_pw [Co][C] (TFLite OHWI with H = W = 1), bias [Co] or null, alpha [Co] or null.
// One blob per output-channel share h: stage s = input channels (2 ip, 2 ip + 1) with ip = (s + h * CO / 2) mod C / 2.
void wide_pack_consts(int C, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst) {
    using K = WK<12, 2>;
    std::fill(dst, dst + wide_consts_floats(C), 0.f);
    for (int h = 0; h < 2; h++) {
        float* blob = dst + (size_t)h * K::TOTAL;
        for (int st = 0; st < K::NST; st++) {
            const int ip = (st + h * K::CO / 2) % (C / 2);
            float* rec = blob + (size_t)st * K::ST_F;
            for (int t = 0; t < 9; t++)
                for (int e = 0; e < 2; e++) rec[2 * t + e] = w_dw[t * C + 2 * ip + e];
            for (int half = 0; half < 2; half++)
                for (int o = 0; o < K::CO; o++) rec[32 + half * K::CO + o] = w_pw[(size_t)(h * K::CO + o) * C + 2 * ip + half];
        }
        float* pb = blob + K::OFF_BIAS;
        float* ps = blob + K::OFF_SLOPE;
        for (int o = 0; o < K::CO; o++) {
            const int c = h * K::CO + o;
            double acc = bias ? bias[c] : 0.0;
            if (b_dw)
                for (int k = 0; k < C; k++) acc += (double)w_pw[(size_t)c * C + k] * b_dw[k];
            pb[o] = (float)acc;
            ps[o] = act == ACT_PRELU ? alpha[c] : (act == ACT_NONE ? 1.f : 0.f);
        }
    }
}

const char* wide_kernel_label(const BlockArgs& a, char* buf, size_t cap) {
    snprintf(buf, cap, "wide_kernel<%d,2,%d>", a.C / 4, a.ep.act == ACT_RELU ? 1 : 0);
    return buf;
}

int launch_wide(const BlockArgs& a, void* stream) {
    WideArgs wa;
    wa.in = a.in; wa.out = a.out; wa.consts = a.w_strip; wa.in_fs = a.in_fs; wa.out_fs = a.out_fs;
    wa.B = a.B; wa.H = a.H;
    wa.pairs = (a.B + 1) / 2;
    // bands: about four workgroups (eight waves) per CU over the chip; a band costs two priming rows of depthwise work
    static const int forced = getenv("MI_WIDE_BAND") ? atoi(getenv("MI_WIDE_BAND")) : 0;  // tuning aid
    static const int nbuf = getenv("MI_WIDE_NBUF") ? atoi(getenv("MI_WIDE_NBUF")) : 2;    // tuning aid
    long bands = std::max<long>(1, (4L * device_cu_count() + wa.pairs / 2) / wa.pairs);
    int rows = (int)((a.H + bands - 1) / bands);
    rows = std::max(rows, std::min(a.H, 4));
    if (forced > 0) rows = std::min(forced, a.H);
    wa.band_rows = rows;
    wa.bands = (a.H + rows - 1) / rows;
    wa.has_res = a.ep.res_mode == RES_DIRECT;
    wa.hi = a.ep.act == ACT_RELU6 ? 6.f : INFINITY;
    hipStream_t s = (hipStream_t)stream;
    const bool relu = a.ep.act == ACT_RELU;
    if (nbuf == 3) return relu ? launch_wide_inst<12, 2, true, 3>(wa, s) : launch_wide_inst<12, 2, false, 3>(wa, s);
    return relu ? launch_wide_inst<12, 2, true, 2>(wa, s) : launch_wide_inst<12, 2, false, 2>(wa, s);
}

}  // namespace mi
