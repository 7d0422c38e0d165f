// mstrip_kernels.hip — fused BlazeBlocks for 32-pixel-wide layers with 48 channels (BackCamera's seven 32x32x48 blocks), the
// pointwise conv on the matrix cores, everything else in the matrix operand layout:
//
//   out = act( PW1x1( DW3x3(in) + b_dw ) + b_pw + in )
//
// Same operator chain as strip_kernels.hip / block_kernels.hip (DEPTHWISE_CONV_2D -> CONV_2D 1x1 -> ADD -> RELU behind
// `interpreter.invoke()`, /root/reference/src/face_detection_lite/face_detection.rs:235; graph: SURVEY.md Appendix A.1,
// "7 x Block(48->48, s1)").  Why not the strip kernel's way (lane = pixel, weights as SGPR operands of v_pk_fma_f32): measured in
// round 3 (tools/wide_bench.hip history, tools/valu_bench.hip) —
//   * a stage's weights (18 + 2 x 48 floats) exceed the SGPR file, so the output channels had to be split over two waves that
//     each repeat the depthwise stage; every wave then streams 6.3 KB of constants through the scalar cache per 64-pixel row
//     and parks ~170 cycles at each of its 48 waits per row (8 waves per CU: SQ_WAIT_ANY 59 %): 43 us per block, no faster
//     than the block kernel;
//   * v_pk_fma_f32 issues at 5.0 cycles per wave alone on a SIMD and 3.47 with two waves (104 / 116 TFLOP/s chip-wide), while
//     v_mfma_f32_16x16x4_f32 sustains its 32 cycles (142-145 TFLOP/s), and a second wave's packed FMAs overlap it almost
//     entirely (MFMA + 1 pk_fma per wave: 34.2 cycles per pair at two waves per SIMD, 46.4 at one).
// So here:
//   * one WAVE = one frame's 32-pixel rows, walking down a band of rows; 48 -> 48 on a 32-pixel row is M = 48 (3 tiles) x
//     N = 32 (2 tiles) x K = 48 (12 steps) of v_mfma_f32_16x16x4_f32 with NO padding: 72 MFMAs = 2304 cycles per row.
//   * the lane layout is the MFMA B-operand's from the start: lane (kq = lane / 16, p = lane % 16) owns channel 4 ks + kq of
//     pixels p and 16 + p for every k-step ks.  The depthwise 3x3 runs in that layout (vertical reuse in registers: the two
//     unfinished output rows an input row contributes to), so its result IS the B operand: no transposition anywhere.  Its taps
//     (9 x 12 values per lane) live in VGPRs for the whole kernel; packed FMAs pair the two pixel tiles.
//   * the pointwise weights are the A operands: 36 wave-wide registers' worth, read from LDS (one copy per workgroup, loaded
//     once), never streamed.  No scalar-cache traffic in the row loop at all.
//   * the result tile has 4 consecutive output channels of one pixel per lane: bias + skip (the centre row, read from the row
//     image as float4) + ReLU, then one global_store_dwordx4 per tile.
// Input rows arrive by LDS-DMA in wave-private images ([34 pixels][52 floats]: zero border columns, 13th float4 slot unused).
// Exact f32; results match the block kernel to reassociation of the sums and the folded depthwise bias.
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

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MstripArgs {
    const float* in;
    float* out;
    const float* consts;   // mstrip_pack_consts()
    long in_fs, out_fs;
    int B, H;              // W == 32
    int bands, band_rows;
    int has_res;
    float hi;              // upper clamp of the activation (6 for ReLU6, +inf otherwise)
    unsigned long long* stamps;  // diagnostic builds only (MI_MSTRIP_STAMPS): 8 accumulators per wave
};

// A run of such blocks in ONE launch (round 5): a workgroup of eight waves owns a frame, wave w its rows [4 w, 4 w + 4) (32 rows), and
// walks through the blocks with a workgroup barrier between them; block j + 1 reads what block j wrote to global memory (every
// intermediate tensor keeps its place in the arena; the halo rows come from the neighbouring waves' stores: same CU, same L1).
constexpr int kMstripChain = 8;
struct MstripChainArgs {
    const float* in[kMstripChain];
    float* out[kMstripChain];
    const float* consts[kMstripChain];
    long in_fs[kMstripChain], out_fs[kMstripChain];
    int has_res[kMstripChain];
    float hi[kMstripChain];
    int nblocks, B, H;
};

template <int CK>   // k-steps: C = 4 * CK channels in and out (CK % 4 == 0: whole 16-channel output tiles)
struct MK {
    static constexpr int C = 4 * CK, MT = C / 16;
    static constexpr int QP = CK + 1;          // float4 slots per pixel of the row image
    static constexpr int PS = 4 * QP;          // pixel stride (floats)
    static constexpr int IMG_F = 34 * PS;      // one row image
    static constexpr int DPX = 64 / QP;        // pixels one LDS-DMA instruction brings in
    static constexpr int NLD = 32 / DPX;       // LDS-DMA instructions per row
    static_assert(32 % DPX == 0 && NLD == 8, "eight DMA instructions per row (two source bases, four immediate offsets each)");
    static_assert((16 * PS) % 64 == 0 && 16 * PS / 64 < 256, "ds_read2st64_b32 reaches the second pixel tile");
    // constants blob (floats): A operands [CK][MT][64] | taps [CK][4][12] | bias [C] | slopes [C]
    static constexpr int OFF_A = 0, A_F = CK * MT * 64, OFF_TAP = A_F, TAP_F = CK * 48, OFF_BIAS = OFF_TAP + TAP_F, OFF_SLOPE = OFF_BIAS + C, TOTAL = OFF_SLOPE + C;
    static constexpr int WG_F = A_F + 2 * C + TAP_F;   // LDS floats shared by a workgroup: A operands, bias, slopes, taps (read once)
    static constexpr int NBUF = 2;
};

template <int N>
__device__ __forceinline__ void mwait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70); }
__device__ __forceinline__ void mwave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class F, int... KS>
__device__ __forceinline__ void mfor_each(F&& f, std::integer_sequence<int, KS...>) { (f(std::integral_constant<int, KS>{}), ...); }

template <int CK, bool RELU>
__global__ __launch_bounds__(256, 2) void mstrip_kernel(MstripArgs a) {
    using K = MK<CK>;
    constexpr int C = K::C, MT = K::MT, PS = K::PS, QP = K::QP, IMG_F = K::IMG_F, NLD = K::NLD, NBUF = K::NBUF;
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef MI_MSTRIP_STAMPS
    unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
#define MI_MSTAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_acc[k] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define MI_MSTAMP(k)
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane >> 4, p = lane & 15;
    float* wgc = lds;                                   // [A_F] A operands, [C] bias, [C] slopes
    float* img = lds + K::WG_F + wave * NBUF * IMG_F;   // this wave's row images
    // unit = (band, frame): the four waves of a workgroup take four consecutive frames of one band; bands of a frame are B
    // units = B / 4 workgroup ids apart (the same XCD when B % 32 == 0: their shared halo rows meet in one L2)
    const int unit = blockIdx.x * 4 + wave;
    const int band = unit / a.B, b = unit - band * a.B;
    const bool active = band < a.bands;  // whole wave; an idle wave still helps with the constants and meets the barrier
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, a.H);
    const float* in = a.in + (long)b * a.in_fs;

    // ---- LDS-DMA of one input row: instruction k brings in pixels [DPX k, DPX k + DPX): lane -> (pixel lane / QP, quad
    // min(lane % QP, CK - 1)), lanes >= DPX * QP idle; the immediate offset moves source and destination alike, M0 makes up the
    // difference between the image's pixel stride and the tensor's
    const int goff = ((lane / QP) * C + 4 * min(lane % QP, CK - 1)) * 4;
    const unsigned lds_img = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)img);
    auto issue_row = [&](int r, int bi) {
        const char* src = reinterpret_cast<const char*>(in + (long)min(max(r, 0), a.H - 1) * 32 * C);
        const char* src1 = src + (NLD / 2) * K::DPX * C * 4;
        const unsigned dstb = lds_img + (unsigned)((bi * IMG_F + PS) * 4);
        constexpr int ACTIVE = K::DPX * QP;
        static_assert(ACTIVE > 32 && ACTIVE < 64, "exec mask written as two 32-bit halves");
        unsigned long long saved;
#define MI_MS_DMA(base, k) "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, " base " offset:" #k "*%8\n\t"
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, %6\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     MI_MS_DMA("%1", 1) MI_MS_DMA("%1", 2) MI_MS_DMA("%1", 3)
                     "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %2\n\t"
                     MI_MS_DMA("%2", 1) MI_MS_DMA("%2", 2) MI_MS_DMA("%2", 3)
                     "s_mov_b64 exec, %0"
                     : "=&s"(saved)
                     : "s"(src), "s"(src1), "v"(goff), "s"(dstb), "n"((NLD / 2) * K::DPX * PS * 4 - (NLD / 2 - 1) * K::DPX * (PS - C) * 4),
                       "n"((1u << (ACTIVE - 32)) - 1), "n"(K::DPX * (PS - C) * 4), "n"(K::DPX * C * 4)
                     : "memory", "scc", "m0");
#undef MI_MS_DMA
    };
    // rows outside the image are zero padding: the landed (clamped) row is cleared before it is read
    auto fix_row = [&](int r, int bi) {
        if (r >= 0 && r < a.H) return;  // wave-uniform
        float zz = 0.f;
        asm volatile("" : "+v"(zz));
        const float4 z = make_float4(zz, zz, zz, zz);
        float* part = img + bi * IMG_F + PS;
        constexpr int N4 = 32 * QP;
#pragma unroll
        for (int k = 0; k < (N4 + 63) / 64; k++)
            if (64 * (k + 1) <= N4 || lane < N4 - 64 * k) *reinterpret_cast<float4*>(part + 4 * (lane + 64 * k)) = z;
    };
    // border pixel columns (left of x = 0, right of x = 31) are never written by the DMA: cleared once
    if (lane < 2 * QP) {
        const int col = lane / QP, qd = lane - col * QP;
#pragma unroll
        for (int bi = 0; bi < NBUF; bi++) *reinterpret_cast<float4*>(img + bi * IMG_F + col * 33 * PS + 4 * qd) = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // the first row is on its way while the constants below are fetched
    if (active) issue_row(y0 - 1, 0);
    MI_MSTAMP(5)
    // ---- workgroup constants into LDS (the only workgroup-level synchronisation of the kernel)
    for (int i = threadIdx.x; i < K::A_F / 4; i += 256) reinterpret_cast<float4*>(wgc)[i] = reinterpret_cast<const float4*>(a.consts + K::OFF_A)[i];
    for (int i = threadIdx.x; i < 2 * C; i += 256) wgc[K::A_F + i] = a.consts[K::OFF_BIAS + i];
    for (int i = threadIdx.x; i < K::TAP_F / 4; i += 256) reinterpret_cast<float4*>(wgc + K::A_F + 2 * C)[i] = reinterpret_cast<const float4*>(a.consts + K::OFF_TAP)[i];
    // The second row goes out behind the constants, and only it may still be in flight when the row loop starts (vector-memory
    // operations retire in issue order): the compiler counts only its own loads when it places vmcnt waits — always safe, since a
    // counted wait also covers everything older — but the loop's own counted waits assume that nothing else is outstanding.
    if (active) issue_row(y0, 1);
    if (active) mwait_vm<NLD>(); else mwait_vm<0>();
    MI_MSTAMP(6)
    // raw barrier behind an LDS-only wait: __syncthreads() would also drain vmcnt, i.e. wait for the second row
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // depthwise taps of this lane's channels: tap[ks][t] = w_dw[t][4 ks + kq], resident in registers for the whole kernel
    float tap[CK][9];
    {
        const float4* tp = reinterpret_cast<const float4*>(wgc + K::A_F + 2 * C) + kq * 3;
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
    if (!active) return;
    MI_MSTAMP(0)
    // ---- per-lane addresses
    const float* xme = img + p * PS + kq;                       // B layout: left neighbour (image pixel p = x - 1 + 1) of pixel p, channel kq
    const float* sme = img + (1 + p) * PS + 4 * kq;             // D layout: centre pixel p, channels 4 kq .. 4 kq + 3 of a 16-channel tile
    const float* ame = wgc + lane;                              // A operands: [ks][mt][lane]
    const unsigned ooff = (unsigned)(p * C + 4 * kq) * 4u;      // bytes: pixel p, channels 4 kq.. of tile (mt, nt) at + (16 nt * C + 16 mt) * 4

    float accA[CK][2], accB[CK][2];   // partial depthwise rows [k-step][pixel tile]: roles alternate from row to row
    f32x4 D[MT][2];
#pragma unroll
    for (int ks = 0; ks < CK; ks++) accA[ks][0] = accA[ks][1] = accB[ks][0] = accB[ks][1] = 0.f;

    // D <- bias + skip for the output row whose centre input row sits in image bi (called at the end of that row's step)
    auto init_D = [&](int bi) {
        float4 bs[MT], x[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bs[mt] = *reinterpret_cast<const float4*>(wgc + K::A_F + 16 * mt + 4 * kq);
        if (a.has_res) {  // wave-uniform; one branch around all six reads (per tile, each read would be waited for on its own)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) x[mt][nt] = *reinterpret_cast<const float4*>(sme + bi * IMG_F + 16 * nt * PS + 16 * mt);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) D[mt][nt] = f32x4{x[mt][nt].x + bs[mt].x, x[mt][nt].y + bs[mt].y, x[mt][nt].z + bs[mt].z, x[mt][nt].w + bs[mt].w};
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) D[mt][0] = D[mt][1] = f32x4{bs[mt].x, bs[mt].y, bs[mt].z, bs[mt].w};
        }
    };
    typedef __attribute__((address_space(1))) char gchar;
    typedef __attribute__((address_space(1))) f32x4 gf32x4;
    auto epilogue = [&](int y) {
        gchar* dst = (gchar*)(a.out + (long)b * a.out_fs + (long)y * 32 * C);
        asm volatile("" : "+s"(dst));
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            float4 sl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!RELU) sl = *reinterpret_cast<const float4*>(wgc + K::A_F + C + 16 * mt + 4 * kq);
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                f32x4 v = D[mt][nt];
                if (RELU) {
                    v = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                } else {  // act(v) = min(max(v,0) + slope*min(v,0), hi): ReLU (slope 0), PReLU (alpha), none (1), ReLU6 (hi = 6)
                    v = f32x4{fminf(fmaxf(v.x, 0.f) + sl.x * fminf(v.x, 0.f), a.hi), fminf(fmaxf(v.y, 0.f) + sl.y * fminf(v.y, 0.f), a.hi),
                              fminf(fmaxf(v.z, 0.f) + sl.z * fminf(v.z, 0.f), a.hi), fminf(fmaxf(v.w, 0.f) + sl.w * fminf(v.w, 0.f), a.hi)};
                }
                *(gf32x4*)(dst + ooff + (unsigned)((16 * nt * C + 16 * mt) * 4)) = v;
            }
        }
    };

    // One input row r (image bi): its ky = 2 / 1 / 0 taps go to the partial output rows r-1 / r / r+1 (aPN on entry / aC / aPN on
    // exit); with EMIT the finished depthwise row r-1 is the B operand of this row's MFMAs.
    // The LDS reads of the row loop are inline asm with their own waits: left to the compiler, the reads of a whole row are merged
    // across k-steps and hoisted to the top of the row (72 + 36 values live: the depthwise taps get spilled).  Every wait is
    // lgkmcnt(0): scalar loads (kernel arguments the compiler re-reads) share the counter and return out of order.
    const unsigned x_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)const_cast<float*>(xme);
    const unsigned a_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)const_cast<float*>(ame);
    auto row = [&](auto emit, int bi, float (&aPN)[CK][2], float (&aC)[CK][2]) {
        constexpr bool EMIT = decltype(emit)::value;
        const unsigned xr = x_lds + (unsigned)(bi * IMG_F * 4), ar = a_lds;
        float xs[2][3][2], av[2][MT];
        auto load_ks = [&](auto ksc, float (&x)[3][2], float (&aw)[MT]) {  // k-step ks: the three tap columns of both pixel tiles; the A operands
            constexpr int ks = decltype(ksc)::value;
            const unsigned xa = xr, aa = ar;  // (named unconditionally: a capture used only inside `if constexpr` is lost by clang)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[dx][0]) : "v"(xa), "n"((dx * PS + 4 * ks) * 4));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[dx][1]) : "v"(xa), "n"((dx * PS + 4 * ks + 16 * PS) * 4));
            }
            if constexpr (EMIT) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(aw[mt]) : "v"(aa), "n"((ks * MT + mt) * 64 * 4));
            }
        };
        // the loaded registers pass through an (empty) asm statement behind the wait: their consumers then depend on something that is
        // ordered behind the s_waitcnt (to the compiler the ds_read asm "returned" its value at once)
        auto landed = [&](float (&x)[3][2], float (&aw)[MT]) {
#pragma unroll
            for (int dx = 0; dx < 3; dx++) asm volatile("" : "+v"(x[dx][0]), "+v"(x[dx][1]));
            if constexpr (EMIT) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) asm volatile("" : "+v"(aw[mt]));
            }
        };
        auto kstep = [&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if constexpr (ks + 1 < CK) load_ks(std::integral_constant<int, ks + 1>{}, xs[(ks + 1) & 1], av[(ks + 1) & 1]);
            float (&x)[3][2] = xs[ks & 1];
            float (&aw)[MT] = av[ks & 1];
            const float (&w)[9] = tap[ks];
            float pch[2];
            // plain v_fma_f32 (the tap is one register for both tiles; a packed FMA would need it duplicated into a register pair):
            // 18 per k-step, which the partner wave's MFMAs cover
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                float n = x[0][nt] * w[0], c = aC[ks][nt], pc = aPN[ks][nt];
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    if (EMIT) pc = __builtin_fmaf(x[dx][nt], w[6 + dx], pc);
                    if (dx) n = __builtin_fmaf(x[dx][nt], w[dx], n);
                    c = __builtin_fmaf(x[dx][nt], w[3 + dx], c);
                }
                aC[ks][nt] = c;
                aPN[ks][nt] = n;
                // pinned here: left alone, LLVM sinks these updates into the next row's block (their only use), which keeps every
                // k-step's pixels alive to the end of the row
                asm volatile("" : "+v"(aC[ks][nt]), "+v"(aPN[ks][nt]));
                pch[nt] = pc;
            }
            if constexpr (EMIT) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    D[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt], pch[0], D[mt][0], 0, 0, 0);
                    D[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt], pch[1], D[mt][1], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next k-step's operands (a whole k-step of cover)
            if constexpr (ks + 1 < CK) landed(xs[(ks + 1) & 1], av[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
        };
        load_ks(std::integral_constant<int, 0>{}, xs[0], av[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        landed(xs[0], av[0]);
        __builtin_amdgcn_sched_barrier(0);
        mfor_each(kstep, std::make_integer_sequence<int, CK>{});
    };

    // vmcnt bookkeeping: vector-memory operations retire in issue order.  A step's operations: the stores of its output row (2 MT, from
    // the third step on), then the DMA of the row two steps ahead (NLD, while the band has one).  "Row r has landed" = at most what was
    // issued behind its DMA is outstanding: the previous step's stores and DMA.
    constexpr int NST = 2 * MT;
    auto wait_row = [&](int t) {  // t = step number (row y0 - 1 + t)
        const bool st = t - 1 >= 2, dm = t == 0 || (t - 1) + 2 <= (y1 - y0 + 1);  // step 0: the prologue's second row is behind it
        if (st && dm) mwait_vm<NST + NLD>();
        else if (dm) mwait_vm<NLD>();
        else if (st) mwait_vm<NST>();
        else mwait_vm<0>();
    };
    auto step = [&](auto emit, int r, float (&aPN)[CK][2], float (&aC)[CK][2]) {
        const int t = r - (y0 - 1), bi = t & 1;
        wait_row(t);
        fix_row(r, bi);
        mwave_sync();
        MI_MSTAMP(1)
        row(emit, bi, aPN, aC);
        MI_MSTAMP(2)
        __builtin_amdgcn_sched_barrier(0);  // (fences: the old and the new accumulator tiles are never live together)
        if constexpr (decltype(emit)::value) epilogue(r - 1);
        MI_MSTAMP(3)
        __builtin_amdgcn_sched_barrier(0);
        init_D(bi);     // output row r starts from bias + its skip, the centre pixels of input row r
        __builtin_amdgcn_sched_barrier(0);
        mwave_sync();   // every read of image bi is issued before the DMA below overwrites it
        __builtin_amdgcn_s_waitcnt(0xC07F);  // ... and has returned (LDS-DMA writes are not ordered behind this wave's earlier reads)
        if (r + 2 <= y1) issue_row(r + 2, bi);
        MI_MSTAMP(4)
    };
    step(std::false_type{}, y0 - 1, accA, accB);
    step(std::false_type{}, y0, accB, accA);
    for (int r = y0 + 1; r <= y1; r += 2) {
        step(std::true_type{}, r, accA, accB);
        if (r + 1 > y1) break;
        step(std::true_type{}, r + 1, accB, accA);
    }
#ifdef MI_MSTRIP_STAMPS
    if (a.stamps && lane == 0)
        for (int k = 0; k < 7; k++) a.stamps[(long)unit * 8 + k] = st_acc[k];
#endif
}

template <int CK, bool RELU>
__global__ __launch_bounds__(512, 2) void mstrip_chain_kernel(MstripChainArgs ca) {
    using K = MK<CK>;
    constexpr int C = K::C, MT = K::MT, PS = K::PS, QP = K::QP, IMG_F = K::IMG_F, NLD = K::NLD, NBUF = K::NBUF;
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef MI_MSTRIP_STAMPS
    unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
#define MI_MSTAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_acc[k] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define MI_MSTAMP(k)
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane >> 4, p = lane & 15;
    float* img = lds + 2 * K::WG_F + wave * NBUF * IMG_F;   // this wave's row images (behind the two constant areas)
    const int b = blockIdx.x;                               // frame
    const int y0 = 4 * wave, y1 = y0 + 4;                   // this wave's rows (H == 32)
    struct { int H; } a{ca.H};
    const float* in = nullptr;
    float* wgc = lds;

    // ---- LDS-DMA of one input row: instruction k brings in pixels [DPX k, DPX k + DPX): lane -> (pixel lane / QP, quad
    // min(lane % QP, CK - 1)), lanes >= DPX * QP idle; the immediate offset moves source and destination alike, M0 makes up the
    // difference between the image's pixel stride and the tensor's
    const int goff = ((lane / QP) * C + 4 * min(lane % QP, CK - 1)) * 4;
    const unsigned lds_img = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)img);
    auto issue_row = [&](int r, int bi) {
        const char* src = reinterpret_cast<const char*>(in + (long)min(max(r, 0), a.H - 1) * 32 * C);
        const char* src1 = src + (NLD / 2) * K::DPX * C * 4;
        const unsigned dstb = lds_img + (unsigned)((bi * IMG_F + PS) * 4);
        constexpr int ACTIVE = K::DPX * QP;
        static_assert(ACTIVE > 32 && ACTIVE < 64, "exec mask written as two 32-bit halves");
        unsigned long long saved;
#define MI_MS_DMA(base, k) "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, " base " offset:" #k "*%8\n\t"
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, %6\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     MI_MS_DMA("%1", 1) MI_MS_DMA("%1", 2) MI_MS_DMA("%1", 3)
                     "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %2\n\t"
                     MI_MS_DMA("%2", 1) MI_MS_DMA("%2", 2) MI_MS_DMA("%2", 3)
                     "s_mov_b64 exec, %0"
                     : "=&s"(saved)
                     : "s"(src), "s"(src1), "v"(goff), "s"(dstb), "n"((NLD / 2) * K::DPX * PS * 4 - (NLD / 2 - 1) * K::DPX * (PS - C) * 4),
                       "n"((1u << (ACTIVE - 32)) - 1), "n"(K::DPX * (PS - C) * 4), "n"(K::DPX * C * 4)
                     : "memory", "scc", "m0");
#undef MI_MS_DMA
    };
    // rows outside the image are zero padding: the landed (clamped) row is cleared before it is read
    auto fix_row = [&](int r, int bi) {
        if (r >= 0 && r < a.H) return;  // wave-uniform
        float zz = 0.f;
        asm volatile("" : "+v"(zz));
        const float4 z = make_float4(zz, zz, zz, zz);
        float* part = img + bi * IMG_F + PS;
        constexpr int N4 = 32 * QP;
#pragma unroll
        for (int k = 0; k < (N4 + 63) / 64; k++)
            if (64 * (k + 1) <= N4 || lane < N4 - 64 * k) *reinterpret_cast<float4*>(part + 4 * (lane + 64 * k)) = z;
    };
    // border pixel columns (left of x = 0, right of x = 31) are never written by the DMA: cleared once
    if (lane < 2 * QP) {
        const int col = lane / QP, qd = lane - col * QP;
#pragma unroll
        for (int bi = 0; bi < NBUF; bi++) *reinterpret_cast<float4*>(img + bi * IMG_F + col * 33 * PS + 4 * qd) = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // constants of block 0 into the first area
    for (int i = threadIdx.x; i < K::A_F / 4; i += 512) reinterpret_cast<float4*>(wgc)[i] = reinterpret_cast<const float4*>(ca.consts[0] + K::OFF_A)[i];
    for (int i = threadIdx.x; i < 2 * C; i += 512) wgc[K::A_F + i] = ca.consts[0][K::OFF_BIAS + i];
    for (int i = threadIdx.x; i < K::TAP_F / 4; i += 512) reinterpret_cast<float4*>(wgc + K::A_F + 2 * C)[i] = reinterpret_cast<const float4*>(ca.consts[0] + K::OFF_TAP)[i];
  for (int blk = 0; blk < ca.nblocks; blk++) {
    wgc = lds + (blk & 1) * K::WG_F;
    in = ca.in[blk] + (long)b * ca.in_fs[blk];
    const int has_res = ca.has_res[blk];
    const float hi = ca.hi[blk];
    float* const outp = ca.out[blk] + (long)b * ca.out_fs[blk];
    issue_row(y0 - 1, 0);
    // the next block's constants: loads issued now (a few per thread), stored into the other area before this block's closing barrier
    constexpr int NPRE = (K::WG_F / 4 + 511) / 512;
    float4 pre[NPRE];
    const bool more = blk + 1 < ca.nblocks;
    if (more) {
        const float* cn = ca.consts[blk + 1];
#pragma unroll
        for (int k = 0; k < NPRE; k++) {
            const int i = threadIdx.x + 512 * k;   // area layout: [A_F | bias, slopes 2 C | taps]; blob: [A | taps | bias | slopes]
            const int src = i < K::A_F / 4 ? K::OFF_A + 4 * i : (i < (K::A_F + 2 * C) / 4 ? K::OFF_BIAS + 4 * i - K::A_F : K::OFF_TAP + 4 * i - K::A_F - 2 * C);
            pre[k] = i < K::WG_F / 4 ? *reinterpret_cast<const float4*>(cn + src) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    issue_row(y0, 1);
    mwait_vm<NLD>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (blk == 0) {   // block 0's constants (plain stores above) are visible to every wave
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // depthwise taps of this lane's channels: tap[ks][t] = w_dw[t][4 ks + kq], resident in registers for the whole block
    float tap[CK][9];
    {
        const float4* tp = reinterpret_cast<const float4*>(wgc + K::A_F + 2 * C) + kq * 3;
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
    MI_MSTAMP(0)
    // ---- per-lane addresses
    const float* xme = img + p * PS + kq;                       // B layout: left neighbour (image pixel p = x - 1 + 1) of pixel p, channel kq
    const float* sme = img + (1 + p) * PS + 4 * kq;             // D layout: centre pixel p, channels 4 kq .. 4 kq + 3 of a 16-channel tile
    const float* ame = wgc + lane;                              // A operands: [ks][mt][lane]
    const unsigned ooff = (unsigned)(p * C + 4 * kq) * 4u;      // bytes: pixel p, channels 4 kq.. of tile (mt, nt) at + (16 nt * C + 16 mt) * 4

    float accA[CK][2], accB[CK][2];   // partial depthwise rows [k-step][pixel tile]: roles alternate from row to row
    f32x4 D[MT][2];
#pragma unroll
    for (int ks = 0; ks < CK; ks++) accA[ks][0] = accA[ks][1] = accB[ks][0] = accB[ks][1] = 0.f;

    // D <- bias + skip for the output row whose centre input row sits in image bi (called at the end of that row's step)
    auto init_D = [&](int bi) {
        float4 bs[MT], x[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bs[mt] = *reinterpret_cast<const float4*>(wgc + K::A_F + 16 * mt + 4 * kq);
        if (has_res) {  // wave-uniform; one branch around all six reads (per tile, each read would be waited for on its own)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) x[mt][nt] = *reinterpret_cast<const float4*>(sme + bi * IMG_F + 16 * nt * PS + 16 * mt);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) D[mt][nt] = f32x4{x[mt][nt].x + bs[mt].x, x[mt][nt].y + bs[mt].y, x[mt][nt].z + bs[mt].z, x[mt][nt].w + bs[mt].w};
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) D[mt][0] = D[mt][1] = f32x4{bs[mt].x, bs[mt].y, bs[mt].z, bs[mt].w};
        }
    };
    typedef __attribute__((address_space(1))) char gchar;
    typedef __attribute__((address_space(1))) f32x4 gf32x4;
    auto epilogue = [&](int y) {
        gchar* dst = (gchar*)(outp + (long)y * 32 * C);
        asm volatile("" : "+s"(dst));
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            float4 sl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!RELU) sl = *reinterpret_cast<const float4*>(wgc + K::A_F + C + 16 * mt + 4 * kq);
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                f32x4 v = D[mt][nt];
                if (RELU) {
                    v = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                } else {  // act(v) = min(max(v,0) + slope*min(v,0), hi): ReLU (slope 0), PReLU (alpha), none (1), ReLU6 (hi = 6)
                    v = f32x4{fminf(fmaxf(v.x, 0.f) + sl.x * fminf(v.x, 0.f), hi), fminf(fmaxf(v.y, 0.f) + sl.y * fminf(v.y, 0.f), hi),
                              fminf(fmaxf(v.z, 0.f) + sl.z * fminf(v.z, 0.f), hi), fminf(fmaxf(v.w, 0.f) + sl.w * fminf(v.w, 0.f), hi)};
                }
                *(gf32x4*)(dst + ooff + (unsigned)((16 * nt * C + 16 * mt) * 4)) = v;
            }
        }
    };

    // One input row r (image bi): its ky = 2 / 1 / 0 taps go to the partial output rows r-1 / r / r+1 (aPN on entry / aC / aPN on
    // exit); with EMIT the finished depthwise row r-1 is the B operand of this row's MFMAs.
    // The LDS reads of the row loop are inline asm with their own waits: left to the compiler, the reads of a whole row are merged
    // across k-steps and hoisted to the top of the row (72 + 36 values live: the depthwise taps get spilled).  Every wait is
    // lgkmcnt(0): scalar loads (kernel arguments the compiler re-reads) share the counter and return out of order.
    const unsigned x_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)const_cast<float*>(xme);
    const unsigned a_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)const_cast<float*>(ame);
    auto row = [&](auto emit, int bi, float (&aPN)[CK][2], float (&aC)[CK][2]) {
        constexpr bool EMIT = decltype(emit)::value;
        const unsigned xr = x_lds + (unsigned)(bi * IMG_F * 4), ar = a_lds;
        float xs[2][3][2], av[2][MT];
        auto load_ks = [&](auto ksc, float (&x)[3][2], float (&aw)[MT]) {  // k-step ks: the three tap columns of both pixel tiles; the A operands
            constexpr int ks = decltype(ksc)::value;
            const unsigned xa = xr, aa = ar;  // (named unconditionally: a capture used only inside `if constexpr` is lost by clang)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[dx][0]) : "v"(xa), "n"((dx * PS + 4 * ks) * 4));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[dx][1]) : "v"(xa), "n"((dx * PS + 4 * ks + 16 * PS) * 4));
            }
            if constexpr (EMIT) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(aw[mt]) : "v"(aa), "n"((ks * MT + mt) * 64 * 4));
            }
        };
        // the loaded registers pass through an (empty) asm statement behind the wait: their consumers then depend on something that is
        // ordered behind the s_waitcnt (to the compiler the ds_read asm "returned" its value at once)
        auto landed = [&](float (&x)[3][2], float (&aw)[MT]) {
#pragma unroll
            for (int dx = 0; dx < 3; dx++) asm volatile("" : "+v"(x[dx][0]), "+v"(x[dx][1]));
            if constexpr (EMIT) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) asm volatile("" : "+v"(aw[mt]));
            }
        };
        auto kstep = [&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if constexpr (ks + 1 < CK) load_ks(std::integral_constant<int, ks + 1>{}, xs[(ks + 1) & 1], av[(ks + 1) & 1]);
            float (&x)[3][2] = xs[ks & 1];
            float (&aw)[MT] = av[ks & 1];
            const float (&w)[9] = tap[ks];
            float pch[2];
            // plain v_fma_f32 (the tap is one register for both tiles; a packed FMA would need it duplicated into a register pair):
            // 18 per k-step, which the partner wave's MFMAs cover
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                float n = x[0][nt] * w[0], c = aC[ks][nt], pc = aPN[ks][nt];
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    if (EMIT) pc = __builtin_fmaf(x[dx][nt], w[6 + dx], pc);
                    if (dx) n = __builtin_fmaf(x[dx][nt], w[dx], n);
                    c = __builtin_fmaf(x[dx][nt], w[3 + dx], c);
                }
                aC[ks][nt] = c;
                aPN[ks][nt] = n;
                // pinned here: left alone, LLVM sinks these updates into the next row's block (their only use), which keeps every
                // k-step's pixels alive to the end of the row
                asm volatile("" : "+v"(aC[ks][nt]), "+v"(aPN[ks][nt]));
                pch[nt] = pc;
            }
            if constexpr (EMIT) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    D[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt], pch[0], D[mt][0], 0, 0, 0);
                    D[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt], pch[1], D[mt][1], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next k-step's operands (a whole k-step of cover)
            if constexpr (ks + 1 < CK) landed(xs[(ks + 1) & 1], av[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
        };
        load_ks(std::integral_constant<int, 0>{}, xs[0], av[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        landed(xs[0], av[0]);
        __builtin_amdgcn_sched_barrier(0);
        mfor_each(kstep, std::make_integer_sequence<int, CK>{});
    };

    // vmcnt bookkeeping: vector-memory operations retire in issue order.  A step's operations: the stores of its output row (2 MT, from
    // the third step on), then the DMA of the row two steps ahead (NLD, while the band has one).  "Row r has landed" = at most what was
    // issued behind its DMA is outstanding: the previous step's stores and DMA.
    constexpr int NST = 2 * MT;
    auto wait_row = [&](int t) {  // t = step number (row y0 - 1 + t)
        const bool st = t - 1 >= 2, dm = t == 0 || (t - 1) + 2 <= (y1 - y0 + 1);  // step 0: the prologue's second row is behind it
        if (st && dm) mwait_vm<NST + NLD>();
        else if (dm) mwait_vm<NLD>();
        else if (st) mwait_vm<NST>();
        else mwait_vm<0>();
    };
    auto step = [&](auto emit, int r, float (&aPN)[CK][2], float (&aC)[CK][2]) {
        const int t = r - (y0 - 1), bi = t & 1;
        wait_row(t);
        fix_row(r, bi);
        mwave_sync();
        MI_MSTAMP(1)
        row(emit, bi, aPN, aC);
        MI_MSTAMP(2)
        __builtin_amdgcn_sched_barrier(0);  // (fences: the old and the new accumulator tiles are never live together)
        if constexpr (decltype(emit)::value) epilogue(r - 1);
        MI_MSTAMP(3)
        __builtin_amdgcn_sched_barrier(0);
        init_D(bi);     // output row r starts from bias + its skip, the centre pixels of input row r
        __builtin_amdgcn_sched_barrier(0);
        mwave_sync();   // every read of image bi is issued before the DMA below overwrites it
        __builtin_amdgcn_s_waitcnt(0xC07F);  // ... and has returned (LDS-DMA writes are not ordered behind this wave's earlier reads)
        if (r + 2 <= y1) issue_row(r + 2, bi);
        MI_MSTAMP(4)
    };
    step(std::false_type{}, y0 - 1, accA, accB);
    step(std::false_type{}, y0, accB, accA);
    for (int r = y0 + 1; r <= y1; r += 2) {
        step(std::true_type{}, r, accA, accB);
        if (r + 1 > y1) break;
        step(std::true_type{}, r + 1, accB, accA);
    }
    // this wave's rows of the block are stored, the next block's constants go into the other area (free since the previous block's
    // closing barrier): then every wave of the frame moves on (vmcnt(0): the stores have left this wave; the waves of a workgroup
    // share the CU's L1, so the neighbours' loads see them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (more) {
        float* nx = lds + ((blk + 1) & 1) * K::WG_F;
#pragma unroll
        for (int k = 0; k < NPRE; k++) {
            const int i = threadIdx.x + 512 * k;
            if (i < K::WG_F / 4) reinterpret_cast<float4*>(nx)[i] = pre[k];
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
}

}  // namespace

// Shapes the kernel takes: stride-1 depthwise 3x3 (SAME) + pointwise with C = Co = 48 on 32-pixel-wide frames, skip = the block's
// own input (or none), constants packed by mstrip_pack_consts().
bool mstrip_shape_ok(int C, int Co) {
    static const bool off = getenv("MI_NO_MSTRIP") != nullptr;  // tuning aid: fall back to the LDS-ring block kernel
    return !off && C == Co && C == 48;
}

bool mstrip_kernel_supports(const BlockArgs& a) {
    if (!a.w_strip || !a.has_dw || a.sh != 1 || a.sw != 1 || a.pt != 1 || a.pl != 1) return false;
    if (!mstrip_shape_ok(a.C, a.Co) || a.W != 32 || a.H != a.Ho || a.W != a.Wo || a.H < 2) return false;
    if (a.ep.res_mode != RES_NONE) {
        if (a.ep.res_after) return false;
        if (a.ep.res_mode != RES_DIRECT || a.ep.res != a.in || a.ep.res_fs != a.in_fs || a.ep.res_C != a.C) return false;
    }
    // a wave walks its band row by row (six dependent steps at four rows): below about one wave per SIMD over the chip the launch is
    // latency-bound and the block kernel's wider workgroups finish sooner (B = 5: 14.3 against 11 us)
    static const int min_b = getenv("MI_MSTRIP_MIN_B") ? atoi(getenv("MI_MSTRIP_MIN_B")) : 32;
    if (a.B < min_b) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return aligned16(a.in) && aligned16(a.out) && !(a.in_fs & 3) && !(a.out_fs & 3);
}

int mstrip_consts_floats(int C) { return C == 48 ? MK<12>::TOTAL : 0; }

// w_dw [3][3][C], b_dw [C] or null, w_pw [Co][C] (TFLite OHWI with H = W = 1), bias [Co] or null, alpha [Co] or null.
void mstrip_pack_consts(int C, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst) {
    using K = MK<12>;
    std::fill(dst, dst + mstrip_consts_floats(C), 0.f);
    // A operand of v_mfma_f32_16x16x4_f32 for (k-step ks, output tile mt): lane l holds W[16 mt + l % 16][4 ks + l / 16]
    for (int ks = 0; ks < K::C / 4; ks++)
        for (int mt = 0; mt < K::MT; mt++)
            for (int l = 0; l < 64; l++) dst[K::OFF_A + (ks * K::MT + mt) * 64 + l] = w_pw[(size_t)(16 * mt + (l & 15)) * C + 4 * ks + (l >> 4)];
    // taps [ks][kq][12]: the nine taps of channel 4 ks + kq (three float4 loads per k-step)
    for (int ks = 0; ks < K::C / 4; ks++)
        for (int kq = 0; kq < 4; kq++)
            for (int t = 0; t < 9; t++) dst[K::OFF_TAP + (ks * 4 + kq) * 12 + t] = w_dw[t * C + 4 * ks + kq];
    for (int c = 0; c < C; c++) {
        // PW(dw + b_dw) + b_pw = PW(dw) + (W b_dw + b_pw): the depthwise bias is folded into the pointwise bias
        double acc = bias ? bias[c] : 0.0;
        if (b_dw)
            for (int k = 0; k < C; k++) acc += (double)w_pw[(size_t)c * C + k] * b_dw[k];
        dst[K::OFF_BIAS + c] = (float)acc;
        dst[K::OFF_SLOPE + c] = act == ACT_PRELU ? alpha[c] : (act == ACT_NONE ? 1.f : 0.f);
    }
}

const char* mstrip_kernel_label(const BlockArgs& a, char* buf, size_t cap) {
    snprintf(buf, cap, "mstrip_kernel<%d,%d>", a.C / 4, a.ep.act == ACT_RELU ? 1 : 0);
    return buf;
}

// A run of blocks the single-block kernel takes, each reading its predecessor's output, on 32 x 32 frames: one launch (see MstripChainArgs)
bool mstrip_chain_supports(const BlockArgs* blocks, int n) {
    static const bool off = getenv("MI_NO_MSTRIP_CHAIN") != nullptr;  // tuning aid: one launch per block
    if (off || n < 2 || n > kMstripChain) return false;
    for (int k = 0; k < n; k++) {
        const BlockArgs& a = blocks[k];
        if (!mstrip_kernel_supports(a) || a.H != 32 || a.B != blocks[0].B || (a.ep.act == ACT_RELU) != (blocks[0].ep.act == ACT_RELU)) return false;
        if (k && a.in != blocks[k - 1].out) return false;
    }
    return true;
}

int launch_mstrip_chain(const BlockArgs* blocks, int n, void* stream) {
    using K = MK<12>;
    if (!mstrip_chain_supports(blocks, n)) return (int)hipErrorInvalidValue;
    MstripChainArgs ca{};
    ca.nblocks = n; ca.B = blocks[0].B; ca.H = blocks[0].H;
    for (int k = 0; k < n; k++) {
        const BlockArgs& a = blocks[k];
        ca.in[k] = a.in; ca.out[k] = a.out; ca.consts[k] = a.w_strip; ca.in_fs[k] = a.in_fs; ca.out_fs[k] = a.out_fs;
        ca.has_res[k] = a.ep.res_mode == RES_DIRECT;
        ca.hi[k] = a.ep.act == ACT_RELU6 ? 6.f : INFINITY;
    }
    const size_t lds_bytes = (size_t)(2 * K::WG_F + 8 * K::NBUF * K::IMG_F) * 4;
    const dim3 grid((unsigned)ca.B);
    hipStream_t s = (hipStream_t)stream;
    if (blocks[0].ep.act == ACT_RELU) {
        auto kern = mstrip_chain_kernel<12, true>;
        if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
        return (int)launch_kernel(kern, grid, dim3(512), lds_bytes, s, ca);
    }
    auto kern = mstrip_chain_kernel<12, false>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    return (int)launch_kernel(kern, grid, dim3(512), lds_bytes, s, ca);
}

unsigned long long* g_mstrip_stamps = nullptr;  // set by the development harness (MI_MSTRIP_STAMPS builds)

int launch_mstrip(const BlockArgs& a, void* stream) {
    using K = MK<12>;
    MstripArgs ma;
    ma.stamps = g_mstrip_stamps;
    ma.in = a.in; ma.out = a.out; ma.consts = a.w_strip; ma.in_fs = a.in_fs; ma.out_fs = a.out_fs;
    ma.B = a.B; ma.H = a.H;
    // bands: about eight waves per CU over the chip; a band costs two priming rows of depthwise work and two halo rows of input
    static const int forced = getenv("MI_MSTRIP_BAND") ? atoi(getenv("MI_MSTRIP_BAND")) : 0;  // tuning aid
    long bands = std::max<long>(1, (8L * device_cu_count() + a.B / 2) / std::max(1, a.B));
    int rows = (int)((a.H + bands - 1) / bands);
    rows = std::max(rows, std::min(a.H, 4));
    if (forced > 0) rows = std::min(forced, a.H);
    ma.band_rows = rows;
    ma.bands = (a.H + rows - 1) / rows;
    ma.has_res = a.ep.res_mode == RES_DIRECT;
    ma.hi = a.ep.act == ACT_RELU6 ? 6.f : INFINITY;
    hipStream_t s = (hipStream_t)stream;
    const long units = (long)a.B * ma.bands;
    const size_t lds_bytes = (size_t)(K::WG_F + 4 * K::NBUF * K::IMG_F) * 4;
    const dim3 grid((unsigned)((units + 3) / 4));
    if (a.ep.act == ACT_RELU) {
        auto kern = mstrip_kernel<12, true>;
        if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
        return (int)launch_kernel(kern, grid, dim3(256), lds_bytes, s, ma);
    }
    auto kern = mstrip_kernel<12, false>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    return (int)launch_kernel(kern, grid, dim3(256), lds_bytes, s, ma);
}

}  // namespace mi
