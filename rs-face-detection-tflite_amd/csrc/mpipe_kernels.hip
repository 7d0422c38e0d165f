// mpipe_kernels.hip — row-pipelined chain of 2..4 same-shape stride-1 BlazeBlocks with 24 channels (BackCamera's 128x128x24 and
// 64x64x24 runs), the pointwise convs on the matrix cores:
//
//   x <- relu( PW1x1( DW3x3(x) + b_dw ) + b_pw + x )     per block, only the first input and the last output in memory
//
// Same operator chains as strip_pipe_kernel (strip_kernels.hip: DEPTHWISE_CONV_2D -> CONV_2D 1x1 -> ADD -> RELU behind
// `interpreter.invoke()`, /root/reference/src/face_detection_lite/face_detection.rs:235; graph: SURVEY.md Appendix A.1), the same
// pipeline — one workgroup = 2 KB waves, waves (2j, 2j+1) run block j on the two 64-column strips of a band (W = 64: on two
// bands), block j's finished row goes through a double-buffered full-width row image in LDS to block j+1, one LDS-only s_barrier
// per row step, block j lags block j-1 by two rows — and mstrip_kernel's arithmetic (mstrip_kernels.hip):
//   * lane (kq = lane / 16, p = lane % 16) owns channel 4 ks + kq of pixels 16 nt + p (nt = 0..3) for every k-step ks: the
//     depthwise 3x3 runs in the MFMA B-operand layout (vertical reuse in registers), its finished row feeds
//     v_mfma_f32_16x16x4_f32 directly; 24 -> 24 on a 64-pixel row is M = 32 (2 tiles, a quarter padding) x N = 64 (4 tiles) x
//     K = 24 (6 steps): 48 MFMAs = 1536 cycles per row against 288 packed FMAs x 3.5 - 5 cycles, and — the point — NO weight
//     stream: the block's depthwise taps (54 registers) and pointwise weights (12 registers of A operands) are resident for the
//     whole launch.  The row pipelines on packed FMAs (strip_pipe2_kernel) spend 41 % of their wave cycles parked at
//     scalar-cache waits for weights they re-read every row pair (profiles/r02_sq_counters_bench.txt).
//   * a result tile holds 4 consecutive output channels of one pixel per lane: bias + skip (the centre row, read back from the
//     input row image) + ReLU, then straight into the next block's row image (two ds_write_b64) or to memory.
// Row images: block 0's arrive by LDS-DMA ([66 pixels][28 floats], zero / neighbour-strip halo columns); the hand-over rings are
// [132 pixels][26 floats]: a 26-float pixel stride makes the b32 reads of the B layout conflict-free (28: two-way).
// Exact f32; results match strip_pipe_kernel to reassociation of the sums.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "kernels.hpp"
#include "launch.hpp"

namespace mi {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kMaxMpipe = 4;
struct MpipeArgs {
    const float* in;
    float* out;
    const float* consts[kMaxMpipe];  // mpipe_pack_consts() blobs, one per block
    long in_fs, out_fs;
    int B, H, W;
    int strips, bands, band_rows, units;
    int has_res[kMaxMpipe];
    float hi[kMaxMpipe];
};

struct MP {
    static constexpr int CK = 6, C = 24, MT = 2;
    static constexpr int PSD = 28;                 // pixel stride of block 0's DMA images (16-byte slots: 6 quads + 1)
    static constexpr int PSR = 26;                 // pixel stride of the hand-over rings (conflict-free b32 reads, 8-byte aligned pixels)
    static constexpr int IMGD = 66 * PSD;          // floats of one DMA row image (64 pixels + 2 halo)
    static constexpr int IMGR = 132 * PSR;         // floats of one full-width ring row image (<= 130 pixels used)
    static constexpr int QP = 7, DPX = 9, NLD = 8; // DMA: 7 float4 slots per pixel, 9 pixels (63 lanes) per instruction, 8 instructions (the last: 3 pixels)
    // constants blob (floats): A operands [CK][MT][64] | taps [CK][4][12] | bias [32] | slopes [32]
    static constexpr int OFF_A = 0, A_F = CK * MT * 64, OFF_TAP = A_F, TAP_F = CK * 48, OFF_BIAS = OFF_TAP + TAP_F, OFF_SLOPE = OFF_BIAS + 32, TOTAL = OFF_SLOPE + 32;
    static constexpr int CW_F = A_F + 64;          // LDS floats per block: A operands, bias, slopes
};

template <int N>
__device__ __forceinline__ void pwait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70); }
__device__ __forceinline__ void pwave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <class F, int... KS>
__device__ __forceinline__ void pfor_each(F&& f, std::integer_sequence<int, KS...>) { (f(std::integral_constant<int, KS>{}), ...); }

// One input row of a 64-pixel strip in the B layout (see mstrip_kernel's row): xr = LDS byte address of the lane's left neighbour,
// channel kq, in a row image of pixel stride PS floats.  Its ky = 2 / 1 / 0 taps go to the partial output rows r-1 / r / r+1 (aPN on
// entry / aC / aPN on exit); with EMIT the finished depthwise row r-1 is the B operand of 48 MFMAs into D.
template <int PS, bool EMIT>
__device__ __forceinline__ void mpipe_row(unsigned xr, unsigned ar, const float (&tap)[MP::CK][9], float (&aPN)[MP::CK][4], float (&aC)[MP::CK][4],
                                          f32x4 (&D)[MP::MT][4]) {
    constexpr int CK = MP::CK, MT = MP::MT;
    float xs[3][4];  // ONE set: the next k-step's pixels are requested behind this k-step's depthwise FMAs, under its MFMAs
    float aw[2][MT]; // A operands (pointwise weights) of the current / the next k-step, from the block's LDS copy ([ks][mt][lane]: ar = lane's address)
    auto load_a = [&](auto ksc, float (&av)[MT]) {
        constexpr int ks = decltype(ksc)::value;
        const unsigned aa = ar;
        if constexpr (EMIT) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(av[mt]) : "v"(aa), "n"((ks * MT + mt) * 64 * 4));
        }
    };
    auto load_ks = [&](auto ksc, float (&x)[3][4]) {
        constexpr int ks = decltype(ksc)::value;
        const unsigned xa = xr;
#pragma unroll
        for (int dx = 0; dx < 3; dx++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[dx][nt]) : "v"(xa), "n"(((16 * nt + dx) * PS + 4 * ks) * 4));
    };
    auto kstep = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        float (&x)[3][4] = xs;
        const float (&w)[9] = tap[ks];
        float pch[4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            float n = x[0][nt] * w[0], c = aC[ks][nt], pc = aPN[ks][nt];
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                if (EMIT) pc = __builtin_fmaf(x[dx][nt], w[6 + dx], pc);
                if (dx) n = __builtin_fmaf(x[dx][nt], w[dx], n);
                c = __builtin_fmaf(x[dx][nt], w[3 + dx], c);
            }
            aC[ks][nt] = c;
            aPN[ks][nt] = n;
            asm volatile("" : "+v"(aC[ks][nt]), "+v"(aPN[ks][nt]));  // (pinned: or LLVM sinks these into the next row's block)
            pch[nt] = pc;
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ks + 1 < CK) {
            load_ks(std::integral_constant<int, ks + 1>{}, xs);
            load_a(std::integral_constant<int, ks + 1>{}, aw[(ks + 1) & 1]);
        }
        if constexpr (EMIT) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < 4; nt++) D[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[ks & 1][mt], pch[nt], D[mt][nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next k-step's pixels (eight MFMAs of cover)
        __builtin_amdgcn_sched_barrier(0);
    };
    load_ks(std::integral_constant<int, 0>{}, xs);
    load_a(std::integral_constant<int, 0>{}, aw[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    pfor_each(kstep, std::make_integer_sequence<int, CK>{});
}

template <int KB, bool RELU>
__global__ __launch_bounds__(128 * KB, 2) void mpipe_kernel(MpipeArgs a) {
    constexpr int CK = MP::CK, C = MP::C, MT = MP::MT, PSD = MP::PSD, PSR = MP::PSR, IMGD = MP::IMGD, IMGR = MP::IMGR, NLD = MP::NLD;
    constexpr int S = KB, NT = 128 * KB, SLOTS = 2;
    constexpr int RING_F = (KB - 1) * SLOTS * IMGR;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane >> 4, p16 = lane & 15;
    const int p = w & 1, role = w >> 1;
    float* dma = lds + RING_F + p * 2 * IMGD;  // block 0 waves: two row buffers each
    const int unit = a.strips == 2 ? (int)blockIdx.x : 2 * (int)blockIdx.x + p;
    const bool active = unit < a.units;
    const int band = unit % a.bands, b = min(unit / a.bands, a.B - 1);
    const int x0 = a.strips == 2 ? 64 * p : 0;
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, a.H);
    const float* in = a.in + (long)b * a.in_fs;
    const float* cst = a.consts[role];
    const bool has_res = a.has_res[role] != 0;
    const float hi = a.hi[role];
    const int lo_j = y0 - (S - 1 - role), hi_j = y1 + (S - 1 - role);  // this block produces rows [lo_j, hi_j)
    const int img_p = a.strips == 2 ? 64 * PSR * p : 66 * PSR * p;     // this wave's window / sub-image inside a ring row image

    // the hand-over rings are cleared once: their border pixel columns (left of x = 0, right of x = W-1) and pad floats are never written
    for (int i = threadIdx.x; i < RING_F / 2; i += NT) reinterpret_cast<float2*>(lds)[i] = make_float2(0.f, 0.f);

    // ---- constants of this wave's block: the depthwise taps resident in registers, the A operands / bias / slopes in the block's LDS area
    // (each of the block's two waves copies half: 832 floats = 208 float4)
    float* cw = lds + RING_F + 4 * IMGD + role * MP::CW_F;
    for (int i = p * 64 + lane; i < MP::CW_F / 4; i += 128) {
        const int f = 4 * i;  // A [0, A_F) | bias [A_F, A_F + 32) | slopes
        const float* srcp = f < MP::A_F ? cst + MP::OFF_A + f : cst + MP::OFF_BIAS + (f - MP::A_F);
        *reinterpret_cast<float4*>(cw + f) = *reinterpret_cast<const float4*>(srcp);
    }
    float tap[CK][9];
    {
        const float* tp = cst + MP::OFF_TAP + kq * 12;
#pragma unroll
        for (int ks = 0; ks < CK; ks++)
#pragma unroll
            for (int t = 0; t < 9; t++) tap[ks][t] = tp[ks * 48 + t];
    }
    // everything above is in its registers / in LDS before the first LDS-DMA goes out: the compiler counts only its own loads when it
    // places vmcnt waits, and the counted waits of the row loop assume that only the loop's own operations are outstanding
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < CK; ks++)
#pragma unroll
        for (int t = 0; t < 9; t++) asm volatile("" : "+v"(tap[ks][t]));
    const unsigned ar = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(cw + lane);
    const float* cbias = cw + MP::A_F + 4 * kq;  // tile mt at + 16 mt; slopes at + 32

    // ---- block 0: LDS-DMA of one input row into a [66 pixels][PSD] image: instruction k brings in image pixels [9 k, 9 k + 9) (63 lanes:
    // lane -> pixel lane / 7, float4 slot lane % 7, the 7th slot repeats quad 5), the last one 3 pixels.  Image pixel i = input column
    // x0 - 1 + i; the two that may lie outside the row (i = 0 of the left strip, i = 65 of the right one) read a clamped column and are
    // cleared once landed.
    const int gl_px = lane / 7, gl_q = min(lane - 7 * (lane / 7), 5);
    int goff_first, goff_mid, goff_last;
    {
        auto off = [&](int i) { return (min(max(x0 - 1 + i, 0), a.W - 1) * C + 4 * gl_q) * 4; };  // bytes from the row start
        goff_first = off(gl_px);
        goff_mid = (gl_px * C + 4 * gl_q) * 4;                    // relative to the group's first pixel, groups 1..6
        goff_last = off(63 + min(gl_px, 2));
    }
    const unsigned lds_dma = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)dma);
    auto issue_row = [&](int r, int bi) {
        const char* src = reinterpret_cast<const char*>(in + (long)min(max(r, 0), a.H - 1) * a.W * C);
        const char* srcm = src + (long)(x0 - 1 + 9) * C * 4;     // group 1's first pixel
        const unsigned dstb = lds_dma + (unsigned)(bi * IMGD * 4);
        unsigned long long saved;
        // ONE asm statement (exec narrowed around the instructions: nothing the compiler schedules may land in between).  Groups 1..6
        // share one per-lane offset; the immediate offset moves source and destination alike, M0 makes up the difference between the
        // image's pixel-group stride (9 x 112 B) and the tensor's (9 x 96 B)
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, 0x7fffffff\n\t"
                     "s_mov_b32 m0, %5\n\tglobal_load_lds_dwordx4 %2, %1\n\t"                                  // group 0: own offsets, base = row start
                     "s_add_u32 m0, m0, 1008\n\tglobal_load_lds_dwordx4 %3, %6\n\t"                            // group 1
                     "s_add_u32 m0, m0, 144\n\tglobal_load_lds_dwordx4 %3, %6 offset:864\n\t"
                     "s_add_u32 m0, m0, 144\n\tglobal_load_lds_dwordx4 %3, %6 offset:1728\n\t"
                     "s_add_u32 m0, m0, 144\n\tglobal_load_lds_dwordx4 %3, %6 offset:2592\n\t"
                     "s_add_u32 m0, m0, 144\n\tglobal_load_lds_dwordx4 %3, %6 offset:3456\n\t"
                     "s_add_u32 m0, m0, 4464\n\tglobal_load_lds_dwordx4 %3, %7\n\t"                            // group 6: own base (5 x 864 would not fit the offset field), M0 = 6 x 1008
                     "s_mov_b32 exec_lo, 0x1fffff\n\ts_mov_b32 exec_hi, 0\n\t"
                     "s_add_u32 m0, m0, 1008\n\tglobal_load_lds_dwordx4 %4, %1\n\t"                            // group 7: pixels 63..65, own offsets
                     "s_mov_b64 exec, %0"
                     : "=&s"(saved)
                     : "s"(src), "v"(goff_first), "v"(goff_mid), "v"(goff_last), "s"(dstb), "s"(srcm), "s"(srcm + 5 * 864)
                     : "memory", "scc");
    };
    const int zl = x0 == 0 ? 0 : -1;                // image pixel columns to clear (-1: none)
    const int zr = x0 + 64 >= a.W ? 65 : -1;
    const int zfix = lane < 7 ? (zl >= 0 ? zl * PSD + 4 * lane : -1) : (lane < 14 ? (zr >= 0 ? zr * PSD + 4 * (lane - 7) : -1) : -1);
    auto fix_row = [&](int r, int bi) {
        float* buf = dma + bi * IMGD;
        float zz = 0.f;
        asm volatile("" : "+v"(zz));
        const float4 z = make_float4(zz, zz, zz, zz);
        if (r < 0 || r >= a.H) {  // wave-uniform, only at the top / bottom image edge
#pragma unroll
            for (int k = 0; k < (IMGD / 4 + 63) / 64; k++)
                if (64 * (k + 1) <= IMGD / 4 || lane < IMGD / 4 - 64 * k) *reinterpret_cast<float4*>(buf + 4 * (lane + 64 * k)) = z;
        } else if (zfix >= 0) {
            *reinterpret_cast<float4*>(buf + zfix) = z;
        }
    };

    float acc0[CK][4], acc1[CK][4];
    f32x4 D[MT][4];
#pragma unroll
    for (int ks = 0; ks < CK; ks++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc0[ks][nt] = acc1[ks][nt] = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) D[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int c_first = y0 - S - 2 * role;  // row this block would consume at step 0 (it starts at step 3 * role)
    if (role == 0 && active) {
        issue_row(c_first, 0);
        issue_row(c_first + 1, 1);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();

    auto wg_barrier = [&]() {  // raw s_barrier behind an LDS-only wait: block 0's DMA and the last block's stores stay in flight across it
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // D <- bias + skip for the output row whose centre input row is the image at `simg` (pixel stride PS): D layout = 4 consecutive
    // channels of one pixel per lane, read as two float2 (ring pixels are 8-byte aligned)
    auto init_D = [&](const float* simg, int PS) {
        float4 bias[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bias[mt] = *reinterpret_cast<const float4*>(cbias + 16 * mt);
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                if (has_res) {  // wave-uniform
                    const float* sp = simg + (1 + 16 * nt + p16) * PS + 16 * mt + 4 * kq;
                    const float2 lo = *reinterpret_cast<const float2*>(sp), hi2 = *reinterpret_cast<const float2*>(sp + 2);
                    D[mt][nt] = f32x4{lo.x + bias[mt].x, lo.y + bias[mt].y, hi2.x + bias[mt].z, hi2.y + bias[mt].w};
                } else {
                    D[mt][nt] = f32x4{bias[mt].x, bias[mt].y, bias[mt].z, bias[mt].w};
                }
            }
    };
    auto act4 = [&](f32x4 v, const float4& sl) {
        if (RELU) return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
        return f32x4{fminf(fmaxf(v.x, 0.f) + sl.x * fminf(v.x, 0.f), hi), fminf(fmaxf(v.y, 0.f) + sl.y * fminf(v.y, 0.f), hi),
                     fminf(fmaxf(v.z, 0.f) + sl.z * fminf(v.z, 0.f), hi), fminf(fmaxf(v.w, 0.f) + sl.w * fminf(v.w, 0.f), hi)};
    };
    typedef __attribute__((address_space(1))) char gchar;
    typedef __attribute__((address_space(1))) f32x4 gf32x4;
    const unsigned ooff = (unsigned)((x0 + p16) * C + 4 * kq) * 4u;  // bytes: pixel x0 + p16, channels 4 kq..; tile (mt, nt) at + (16 nt C + 16 mt) * 4

    auto step = [&](int t, float (&aPN)[CK][4], float (&aC)[CK][4]) {
        const int c = c_first + t;  // row consumed in this step; the row it completes is c - 1
        const int e = c - 1;
        if (active && c >= lo_j - 1 && c <= hi_j) {
            const bool emit = e >= lo_j && e < hi_j;
            const float* simg;
            if (role == 0) {
                if (c + 1 <= hi_j) pwait_vm<NLD>();
                else pwait_vm<0>();
                fix_row(c, t & 1);
                pwave_sync();
                simg = dma + (t & 1) * IMGD;
                const unsigned xr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)const_cast<float*>(simg + p16 * PSD + kq);
                if (emit) mpipe_row<PSD, true>(xr, ar, tap, aPN, aC, D);
                else mpipe_row<PSD, false>(xr, ar, tap, aPN, aC, D);
            } else {
                simg = lds + ((role - 1) * SLOTS + ((t - 1) & 1)) * IMGR + img_p;
                const unsigned xr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)const_cast<float*>(simg + p16 * PSR + kq);
                if (emit) mpipe_row<PSR, true>(xr, ar, tap, aPN, aC, D);
                else mpipe_row<PSR, false>(xr, ar, tap, aPN, aC, D);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (emit) {
                float4 slope[MT];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) slope[mt] = RELU ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(cbias + 32 + 16 * mt);
                if (role < S - 1) {
                    // hand the row to the next block (its image pixel 1 + x); rows outside the image are that block's zero padding
                    float* dstl = lds + (role * SLOTS + (t & 1)) * IMGR + img_p + (1 + p16) * PSR + 4 * kq;
                    const bool inside = e >= 0 && e < a.H;  // wave-uniform
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        if (mt == 1 && kq >= 2) continue;  // channels 24..31 do not exist
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) {
                            f32x4 v = act4(D[mt][nt], slope[mt]);
                            if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};
                            float* d = dstl + 16 * nt * PSR + 16 * mt;
                            *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
                            *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
                        }
                    }
                } else {
                    gchar* dst = (gchar*)(a.out + (long)b * a.out_fs + (long)e * a.W * C);
                    asm volatile("" : "+s"(dst));
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        if (mt == 1 && kq >= 2) continue;
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) *(gf32x4*)(dst + ooff + (unsigned)((16 * nt * C + 16 * mt) * 4)) = act4(D[mt][nt], slope[mt]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            init_D(simg, role == 0 ? PSD : PSR);  // output row c starts from bias + its skip, the centre pixels of input row c
            if (role == 0 && c + 2 <= hi_j) {
                pwave_sync();
                __builtin_amdgcn_s_waitcnt(0xC07F);  // every read of this row buffer has returned
                issue_row(c + 2, t & 1);
            }
        }
        wg_barrier();
    };
    const int T = a.band_rows + 3 * S - 1;
    for (int t = 0; t < T; t += 2) {
        step(t, acc0, acc1);
        if (t + 1 >= T) break;
        step(t + 1, acc1, acc0);
    }
}

template <int KB, bool RELU>
int launch_mpipe_inst(const MpipeArgs& pa, hipStream_t s) {
    auto kern = mpipe_kernel<KB, RELU>;
    const size_t lds_bytes = (size_t)((KB - 1) * 2 * MP::IMGR + 4 * MP::IMGD + KB * MP::CW_F) * 4;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    const int wgs = pa.strips == 2 ? pa.units : (pa.units + 1) / 2;
    return (int)launch_kernel(kern, dim3((unsigned)wgs), dim3(128 * KB), lds_bytes, s, pa);
}

}  // namespace

bool mpipe_shape_ok(int C, int Co) {
    static const bool off = getenv("MI_NO_MPIPE") != nullptr;  // tuning aid: strip_pipe_kernel takes the chain
    return !off && C == 24 && Co == 24;
}

int mpipe_consts_floats(int C) { return C == 24 ? MP::TOTAL : 0; }

// w_dw [3][3][C], b_dw [C] or null, w_pw [Co][C] (TFLite OHWI with H = W = 1), bias [Co] or null, alpha [Co] or null.
void mpipe_pack_consts(int C, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst) {
    std::fill(dst, dst + mpipe_consts_floats(C), 0.f);
    // A operand of v_mfma_f32_16x16x4_f32 for (k-step ks, output tile mt): lane l holds W[16 mt + l % 16][4 ks + l / 16] (rows >= C: 0)
    for (int ks = 0; ks < MP::CK; ks++)
        for (int mt = 0; mt < MP::MT; mt++)
            for (int l = 0; l < 64; l++) {
                const int o = 16 * mt + (l & 15);
                if (o < C) dst[MP::OFF_A + (ks * MP::MT + mt) * 64 + l] = w_pw[(size_t)o * C + 4 * ks + (l >> 4)];
            }
    for (int ks = 0; ks < MP::CK; ks++)
        for (int kq = 0; kq < 4; kq++)
            for (int t = 0; t < 9; t++) dst[MP::OFF_TAP + (ks * 4 + kq) * 12 + t] = w_dw[t * C + 4 * ks + kq];
    for (int c = 0; c < C; c++) {
        // PW(dw + b_dw) + b_pw = PW(dw) + (W b_dw + b_pw): the depthwise bias is folded into the pointwise bias
        double acc = bias ? bias[c] : 0.0;
        if (b_dw)
            for (int k = 0; k < C; k++) acc += (double)w_pw[(size_t)c * C + k] * b_dw[k];
        dst[MP::OFF_BIAS + c] = (float)acc;
        dst[MP::OFF_SLOPE + c] = act == ACT_PRELU ? alpha[c] : (act == ACT_NONE ? 1.f : 0.f);
    }
}

// Chains the kernel takes: 2..4 stride-1 blocks, C = Co = 24, W in {64, 128}, skip = each block's own input (or none), no stride-2
// tail, constants packed by mpipe_pack_consts() (BlockArgs::w_mfma).
bool mpipe_supports(const BlockArgs* blocks, int n) {
    if (n < 2 || n > kMaxMpipe) return false;
    const BlockArgs& f = blocks[0];
    if (!mpipe_shape_ok(f.C, f.Co) || (f.W != 64 && f.W != 128) || f.H < 2) return false;
    for (int k = 0; k < n; k++) {
        const BlockArgs& t = blocks[k];
        if (!t.w_mfma || !t.has_dw || t.sh != 1 || t.sw != 1 || t.pt != 1 || t.pl != 1) return false;
        if (t.C != f.C || t.Co != f.Co || t.H != f.H || t.W != f.W || t.Ho != f.H || t.Wo != f.W) return false;
        if (t.ep.res_after || (t.ep.res_mode != RES_NONE && (t.ep.res_mode != RES_DIRECT || t.ep.res_C != t.C))) return false;
        if ((t.ep.act == ACT_RELU) != (f.ep.act == ACT_RELU)) return false;
    }
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return aligned16(f.in) && aligned16(blocks[n - 1].out) && !(f.in_fs & 3) && !(blocks[n - 1].out_fs & 3);
}

const char* mpipe_label(const BlockArgs* blocks, int n, char* buf, size_t cap) {
    snprintf(buf, cap, "mpipe_kernel<%d,%d>", n, blocks[0].ep.act == ACT_RELU ? 1 : 0);
    return buf;
}

int launch_mpipe(const BlockArgs* blocks, int n, void* stream) {
    const BlockArgs& a = blocks[0];
    MpipeArgs pa{};
    pa.in = a.in; pa.out = blocks[n - 1].out; pa.in_fs = a.in_fs; pa.out_fs = blocks[n - 1].out_fs;
    pa.B = a.B; pa.H = a.H; pa.W = a.W;
    pa.strips = a.W / 64;
    for (int k = 0; k < n; k++) {
        pa.consts[k] = blocks[k].w_mfma;
        pa.has_res[k] = blocks[k].ep.res_mode != RES_NONE;
        pa.hi[k] = blocks[k].ep.act == ACT_RELU6 ? 6.f : INFINITY;
    }
    // bands: about one resident set of workgroups over the chip, but not so short that the pipeline fill steps dominate
    static const int forced = getenv("MI_MPIPE_BAND") ? atoi(getenv("MI_MPIPE_BAND")) : 0;  // tuning aid
    const long wg_units = pa.strips == 2 ? 1 : 2;
    long bands = std::max<long>(1, ((long)device_cu_count() * wg_units + a.B / 2) / std::max(1, a.B));
    int rows = (int)((a.H + bands - 1) / bands);
    rows = std::max(rows, std::min(a.H, 8 * n));
    if (forced > 0) rows = std::min(forced, a.H);
    pa.band_rows = rows;
    pa.bands = (a.H + rows - 1) / rows;
    pa.units = a.B * pa.bands;
    hipStream_t s = (hipStream_t)stream;
    const bool relu = a.ep.act == ACT_RELU;
#define MI_MPIPE_CASE(KBV) \
    if (n == KBV) return relu ? launch_mpipe_inst<KBV, true>(pa, s) : launch_mpipe_inst<KBV, false>(pa, s);
    MI_MPIPE_CASE(2) MI_MPIPE_CASE(3) MI_MPIPE_CASE(4)
#undef MI_MPIPE_CASE
    return (int)hipErrorInvalidValue;
}

}  // namespace mi
