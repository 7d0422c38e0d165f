// mwalk_kernels.hip — mstrip_kernels.hip's scheme for other layer shapes: one fused BlazeBlock
//
//   out = act( PW1x1( DW3x3(in) + b_dw ) + b_pw + in )
//
// (DEPTHWISE_CONV_2D -> CONV_2D 1x1 -> ADD -> PRELU/RELU behind `interpreter.invoke()`, /root/reference/src/face_detection_lite/
// face_landmark.rs:265; graph: SURVEY.md Appendix A.3, the face mesh's 48x48x32 and 24x24x64 blocks) with one WAVE walking down a band of
// rows of one frame: W <= 16 WT pixels = WT tiles of v_mfma_f32_16x16x4_f32 (the last one may be partly outside the frame), C = 4 CK
// channels in, 16 MT out.  Lane (kq = lane / 16,
// p = lane % 16) owns channel 4 ks + kq of pixel p of every tile: the depthwise 3x3 runs in the MFMA B-operand layout with its taps
// in registers (mrow.hpp), the pointwise weights are A operands read from LDS, the input rows arrive by LDS-DMA in wave-private
// row images, bias + skip + activation on the result tiles, one 16-byte store per lane and tile.  No workgroup synchronisation
// after the constants are in LDS.  The strip kernel (lane = pixel, weights as SGPR operands of packed FMAs) runs this shape at
// 0.105 ms per block for 512 frames (2.9 TB/s); see mstrip_kernels.hip for the measurements behind the scheme.
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

struct MwalkArgs {
    const float* in;
    float* out;
    const float* consts;   // mwalk_pack_consts()
    long in_fs, out_fs;
    int B, H;
    int bands, band_rows;
    int has_res;
    float hi;              // upper clamp of the activation (6 for ReLU6, +inf otherwise)
};

constexpr int mw_pick_dpx(int qp, int w) {  // pixels per LDS-DMA instruction: 8 instructions per row, as many lanes in use as fit
    for (int d = 64 / qp; d >= 1; d--)
        if (w % d == 0 && w / d == 8) return d;
    return 0;
}

// W_: frame width (default 16 WT; less: the last tile is partly outside the frame — its lanes compute on whatever lies behind the row in
// the image and store nothing).  TAPL_: the depthwise taps stay in LDS and are read per k-step (channel counts whose 10 x CK tap
// registers do not fit beside the accumulators).
template <int CK_, int MT_, int WT_, int W_ = 16 * WT_, bool TAPL_ = false>
struct MW {
    static constexpr int CK = CK_, MT = MT_, WT = WT_, C = 4 * CK, Co = 16 * MT, W = W_;
    static constexpr bool TAPL = TAPL_;
    static_assert(W <= 16 * WT && W > 16 * (WT - 1), "WT tiles cover the row");
    static constexpr int QP = CK + 1, PS = 4 * QP;      // float4 slots / floats per pixel of the row image (odd slot count: a tile's pixels on distinct banks)
    static_assert(QP % 2 == 1, "odd number of float4 slots per pixel");
    static constexpr int IMG_F = (W + 2) * PS;
    static constexpr int SLACK_F = (16 * WT - W) * PS;  // the lanes of a partly used last tile read on behind the row: into the next image, or this slack behind the last one
    static constexpr int DPX = mw_pick_dpx(QP, W), NLD = 8, ACTIVE = DPX * QP;
    static_assert(DPX > 0 && ACTIVE > 32 && ACTIVE < 64, "DMA shape: eight instructions per row (two source bases, four immediate offsets each)");
    static_assert(3 * DPX * C * 4 < 4096, "immediate offsets of the DMA instructions");
    // constants blob (floats): A operands [CK][MT][64] | taps [CK][4][12] | bias [Co] | slopes [Co]
    static constexpr int OFF_A = 0, A_F = CK * MT * 64, OFF_TAP = A_F, TAP_F = CK * 48, OFF_BIAS = OFF_TAP + TAP_F, OFF_SLOPE = OFF_BIAS + Co, TOTAL = OFF_SLOPE + Co;
    static constexpr int NBUF = 2, LDS_F = TOTAL + 4 * NBUF * IMG_F + SLACK_F;
};

template <class K, bool RELU>
__global__ __launch_bounds__(256, 2) void mwalk_kernel(MwalkArgs a) {
    constexpr int CK = K::CK, MT = K::MT, WT = K::WT, C = K::C, Co = K::Co, W = K::W, PS = K::PS, QP = K::QP, IMG_F = K::IMG_F, NLD = K::NLD, NBUF = K::NBUF;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane >> 4, p = lane & 15;
    float* wgc = lds;                                   // the constants blob
    float* img = lds + K::TOTAL + wave * NBUF * IMG_F;  // this wave's row images
    // unit = (band, frame): the four waves of a workgroup take four consecutive frames of one band
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
        const char* src = reinterpret_cast<const char*>(in + (long)min(max(r, 0), a.H - 1) * W * C);
        const char* src1 = src + (NLD / 2) * K::DPX * C * 4;
        const unsigned dstb = lds_img + (unsigned)((bi * IMG_F + PS) * 4);
        unsigned long long saved;
#define MI_MW_DMA(base, k) "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, " base " offset:" #k "*%8\n\t"
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, %6\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     MI_MW_DMA("%1", 1) MI_MW_DMA("%1", 2) MI_MW_DMA("%1", 3)
                     "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %2\n\t"
                     MI_MW_DMA("%2", 1) MI_MW_DMA("%2", 2) MI_MW_DMA("%2", 3)
                     "s_mov_b64 exec, %0"
                     : "=&s"(saved)
                     : "s"(src), "s"(src1), "v"(goff), "s"(dstb), "n"((NLD / 2) * K::DPX * PS * 4 - (NLD / 2 - 1) * K::DPX * (PS - C) * 4),
                       "n"((1u << (K::ACTIVE - 32)) - 1), "n"(K::DPX * (PS - C) * 4), "n"(K::DPX * C * 4)
                     : "memory", "scc", "m0");
#undef MI_MW_DMA
    };
    // rows outside the image are zero padding: the landed (clamped) row is cleared before it is read
    auto fix_row = [&](int r, int bi) {
        if (r >= 0 && r < a.H) return;  // wave-uniform
        float zz = 0.f;
        asm volatile("" : "+v"(zz));
        const float4 z = make_float4(zz, zz, zz, zz);
        float* part = img + bi * IMG_F + PS;
        constexpr int N4 = W * QP;
#pragma unroll
        for (int k = 0; k < (N4 + 63) / 64; k++)
            if (64 * (k + 1) <= N4 || lane < N4 - 64 * k) *reinterpret_cast<float4*>(part + 4 * (lane + 64 * k)) = z;
    };
    // border pixel columns (left of x = 0, right of x = W - 1) are never written by the DMA: cleared once
    if (lane < 2 * QP) {
        const int col = lane / QP, qd = lane - col * QP;
#pragma unroll
        for (int bi = 0; bi < NBUF; bi++) *reinterpret_cast<float4*>(img + bi * IMG_F + col * (W + 1) * PS + 4 * qd) = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // the first row is on its way while the constants are fetched; the second goes out behind them, and only it may still be in flight
    // when the row loop starts (vector-memory operations retire in issue order; the loop's counted waits assume nothing else is outstanding)
    if (active) issue_row(y0 - 1, 0);
    for (int i = threadIdx.x; i < K::TOTAL / 4; i += 256) reinterpret_cast<float4*>(wgc)[i] = reinterpret_cast<const float4*>(a.consts)[i];
    if (active) issue_row(y0, 1);
    if (active) dwait_vm<NLD>(); else dwait_vm<0>();
    dwg_barrier();   // the only workgroup-level synchronisation of the kernel
    asm volatile("" ::: "memory");
    // depthwise taps of this lane's channels: tap[ks][t] = w_dw[t][4 ks + kq], resident in registers for the whole kernel
    dv2f tap[K::TAPL ? 1 : CK][5];
    if constexpr (!K::TAPL) dload_taps<CK>(reinterpret_cast<const float4*>(wgc + K::OFF_TAP) + kq * 3, tap);
    if (!active) return;
    // ---- per-lane addresses
    typedef __attribute__((address_space(3))) float lfloat;
    const unsigned x_lds = (unsigned)(uintptr_t)(lfloat*)(img + p * PS + kq);          // B layout: left neighbour of pixel p, channel kq
    const unsigned a_lds = (unsigned)(uintptr_t)(lfloat*)(wgc + K::OFF_A + lane);      // A operands [ks][mt][lane]
    const float* sme = img + (1 + p) * PS + 4 * kq;                                    // D layout: centre pixel p, channels 4 kq .. of a 16-channel tile
    const unsigned ooff = (unsigned)(p * Co + 4 * kq) * 4u;                             // bytes: + (16 nt * Co + 16 mt) * 4 for tile (mt, nt)

    RowAcc<WT> accA[CK], accB[CK];   // partial depthwise rows: roles alternate from row to row
#pragma unroll
    for (int ks = 0; ks < CK; ks++) { accA[ks].clear(); accB[ks].clear(); }
    df32x4 D[MT][WT];
    // D <- bias + skip for the output row whose centre input row sits in image bi
    auto init_D = [&](int bi) {
        float4 bs[MT], x[MT][WT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bs[mt] = *reinterpret_cast<const float4*>(wgc + K::OFF_BIAS + 16 * mt + 4 * kq);
        if (a.has_res) {  // wave-uniform; one branch around all the reads
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < WT; nt++) {
                    x[mt][nt] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (16 * mt < C) x[mt][nt] = *reinterpret_cast<const float4*>(sme + bi * IMG_F + 16 * nt * PS + 16 * mt);
                }
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < WT; nt++) D[mt][nt] = df32x4{x[mt][nt].x + bs[mt].x, x[mt][nt].y + bs[mt].y, x[mt][nt].z + bs[mt].z, x[mt][nt].w + bs[mt].w};
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < WT; nt++) D[mt][nt] = df32x4{bs[mt].x, bs[mt].y, bs[mt].z, bs[mt].w};
        }
    };
    typedef __attribute__((address_space(1))) char gchar;
    typedef __attribute__((address_space(1))) df32x4 gf32x4;
    auto epilogue = [&](int y) {
        gchar* dst = (gchar*)(a.out + (long)b * a.out_fs + (long)y * W * Co);
        asm volatile("" : "+s"(dst));
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            float4 sl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!RELU) sl = *reinterpret_cast<const float4*>(wgc + K::OFF_SLOPE + 16 * mt + 4 * kq);
#pragma unroll
            for (int nt = 0; nt < WT; nt++) {
                df32x4 v = D[mt][nt];
                if (RELU) {
                    v = df32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                } else {  // act(v) = min(max(v,0) + slope*min(v,0), hi): ReLU (slope 0), PReLU (alpha), none (1), ReLU6 (hi = 6)
                    v = df32x4{fminf(fmaxf(v.x, 0.f) + sl.x * fminf(v.x, 0.f), a.hi), fminf(fmaxf(v.y, 0.f) + sl.y * fminf(v.y, 0.f), a.hi),
                               fminf(fmaxf(v.z, 0.f) + sl.z * fminf(v.z, 0.f), a.hi), fminf(fmaxf(v.w, 0.f) + sl.w * fminf(v.w, 0.f), a.hi)};
                }
                if (16 * (nt + 1) <= W || 16 * nt + p < W) *(gf32x4*)(dst + ooff + (unsigned)((16 * nt * Co + 16 * mt) * 4)) = v;   // (a partly used last tile)
            }
        }
    };

    // vmcnt bookkeeping: vector-memory operations retire in issue order.  A step's operations: the stores of its output row (NST, from
    // the third step on), then the DMA of the row two steps ahead (NLD, while the band has one).  "Row r has landed" = at most what was
    // issued behind its DMA is outstanding: the previous step's stores and DMA.
    constexpr int NST = MT * WT;
    auto wait_row = [&](int t) {  // t = step number (row y0 - 1 + t)
        const bool st = t - 1 >= 2, dm = t == 0 || (t - 1) + 2 <= (y1 - y0 + 1);  // step 0: the prologue's second row is behind it
        if (st && dm) dwait_vm<NST + NLD>();
        else if (dm) dwait_vm<NLD>();
        else if (st) dwait_vm<NST>();
        else dwait_vm<0>();
    };
    auto step = [&](auto emit, int r, RowAcc<WT> (&aPN)[CK], RowAcc<WT> (&aC)[CK]) {
        const int t = r - (y0 - 1), bi = t & 1;
        wait_row(t);
        fix_row(r, bi);
        dwave_sync();
        mdb_row<CK, MT, WT, PS, decltype(emit)::value, K::TAPL>(x_lds + (unsigned)(bi * IMG_F * 4), a_lds, tap, aPN, aC, D, reinterpret_cast<const float4*>(wgc + K::OFF_TAP) + kq * 3);
        __builtin_amdgcn_sched_barrier(0);  // (fences: the old and the new accumulator tiles are never live together)
        if constexpr (decltype(emit)::value) epilogue(r - 1);
        __builtin_amdgcn_sched_barrier(0);
        init_D(bi);     // output row r starts from bias + its skip, the centre pixels of input row r
        __builtin_amdgcn_sched_barrier(0);
        dwave_sync();   // every read of image bi is issued before the DMA below overwrites it
        __builtin_amdgcn_s_waitcnt(0xC07F);  // ... and has returned (LDS-DMA writes are not ordered behind this wave's earlier reads)
        if (r + 2 <= y1) issue_row(r + 2, bi);
    };
    step(std::false_type{}, y0 - 1, accA, accB);
    step(std::false_type{}, y0, accB, accA);
    for (int r = y0 + 1; r <= y1; r += 2) {
        step(std::true_type{}, r, accA, accB);
        if (r + 1 > y1) break;
        step(std::true_type{}, r + 1, accB, accA);
    }
}

using MW48 = MW<8, 2, 3>;   // 48 pixels wide, 32 -> 32 channels (face mesh)
using MW24 = MW<16, 4, 2, 24, true>;   // 24 pixels wide (1.5 tiles), 64 -> 64 channels (face mesh), taps in LDS

template <class K>
void mw_pack(const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst) {
    constexpr int C = K::C, Co = K::Co;
    std::fill(dst, dst + K::TOTAL, 0.f);
    // A operand of v_mfma_f32_16x16x4_f32 for (k-step ks, output tile mt): lane l holds W[16 mt + l % 16][4 ks + l / 16]
    for (int ks = 0; ks < K::CK; ks++)
        for (int mt = 0; mt < K::MT; mt++)
            for (int l = 0; l < 64; l++) dst[K::OFF_A + (ks * K::MT + mt) * 64 + l] = w_pw[(size_t)(16 * mt + (l & 15)) * C + 4 * ks + (l >> 4)];
    // taps [ks][kq][12]: the nine taps of channel 4 ks + kq (three float4 loads per k-step)
    for (int ks = 0; ks < K::CK; ks++)
        for (int kq = 0; kq < 4; kq++)
            for (int t = 0; t < 9; t++) dst[K::OFF_TAP + (ks * 4 + kq) * 12 + t] = w_dw[t * C + 4 * ks + kq];
    for (int c = 0; c < Co; c++) {
        // PW(dw + b_dw) + b_pw = PW(dw) + (W b_dw + b_pw): the depthwise bias is folded into the pointwise bias
        double acc = bias ? bias[c] : 0.0;
        if (b_dw)
            for (int k = 0; k < C; k++) acc += (double)w_pw[(size_t)c * C + k] * b_dw[k];
        dst[K::OFF_BIAS + c] = (float)acc;
        dst[K::OFF_SLOPE + c] = act == ACT_PRELU ? alpha[c] : (act == ACT_NONE ? 1.f : 0.f);
    }
}

template <class K>
int mw_launch(const BlockArgs& a, hipStream_t s) {
    MwalkArgs ma;
    ma.in = a.in; ma.out = a.out; ma.consts = a.w_mwalk; ma.in_fs = a.in_fs; ma.out_fs = a.out_fs;
    ma.B = a.B; ma.H = a.H;
    // bands: about eight waves per CU over the chip; a band costs two priming rows of depthwise work and two halo rows of input
    static const int forced = getenv("MI_MWALK_BAND") ? atoi(getenv("MI_MWALK_BAND")) : 0;  // tuning aid
    long bands = std::max<long>(1, (8L * device_cu_count() + a.B / 2) / std::max(1, a.B));
    int rows = (int)((a.H + bands - 1) / bands);
    rows = std::max(rows, std::min(a.H, 4));
    if (forced > 0) rows = std::min(forced, a.H);
    ma.band_rows = rows;
    ma.bands = (a.H + rows - 1) / rows;
    ma.has_res = a.ep.res_mode == RES_DIRECT;
    ma.hi = a.ep.act == ACT_RELU6 ? 6.f : INFINITY;
    const long units = (long)a.B * ma.bands;
    const size_t lds_bytes = (size_t)K::LDS_F * 4;
    const dim3 grid((unsigned)((units + 3) / 4));
    if (a.ep.act == ACT_RELU) {
        auto kern = mwalk_kernel<K, true>;
        if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
        return (int)launch_kernel(kern, grid, dim3(256), lds_bytes, s, ma);
    }
    auto kern = mwalk_kernel<K, false>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    return (int)launch_kernel(kern, grid, dim3(256), lds_bytes, s, ma);
}

}  // namespace

// Shapes the kernel takes: stride-1 depthwise 3x3 (SAME) + pointwise, skip = the block's own input (or none)
static int mw_shape(int W, int C, int Co) {
    static const bool off = getenv("MI_NO_MWALK") != nullptr;  // tuning aid: the strip / block kernels instead
    if (off) return 0;
    if (W == 48 && C == 32 && Co == 32) return 1;
    if (W == 24 && C == 64 && Co == 64) return 2;
    return 0;
}
bool mwalk_shape_ok(int W, int C, int Co) { return mw_shape(W, C, Co) != 0; }

int mwalk_consts_floats(int W, int C, int Co) { return mw_shape(W, C, Co) == 1 ? MW48::TOTAL : (mw_shape(W, C, Co) == 2 ? MW24::TOTAL : 0); }

// w_dw [3][3][C], b_dw [C] or null, w_pw [Co][C] (TFLite OHWI with H = W = 1), bias [Co] or null, alpha [Co] or null.
void mwalk_pack_consts(int W, int C, int Co, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst) {
    if (mw_shape(W, C, Co) == 1) mw_pack<MW48>(w_dw, b_dw, w_pw, bias, alpha, act, dst);
    if (mw_shape(W, C, Co) == 2) mw_pack<MW24>(w_dw, b_dw, w_pw, bias, alpha, act, dst);
}

bool mwalk_kernel_supports(const BlockArgs& a) {
    if (!a.w_mwalk || !a.has_dw || a.sh != 1 || a.sw != 1 || a.pt != 1 || a.pl != 1) return false;
    if (!mwalk_shape_ok(a.W, a.C, a.Co) || a.H != a.Ho || a.W != a.Wo || a.H < 2) return false;
    if (a.ep.res_mode != RES_NONE) {
        if (a.ep.res_after) return false;
        if (a.ep.res_mode != RES_DIRECT || a.ep.res != a.in || a.ep.res_fs != a.in_fs || a.ep.res_C != a.C) return false;
    }
    // a wave walks its band row by row: below about one wave per SIMD over the chip the launch is latency-bound
    // (the 24-pixel rows fill 1.5 of their 2 tiles: 0.0211 against the block kernel's 0.0201 ms at 64 frames, 0.0233 / 0.0322 at 128)
    static const int min_b = getenv("MI_MWALK_MIN_B") ? atoi(getenv("MI_MWALK_MIN_B")) : 0;
    if (a.B < (min_b > 0 ? min_b : (mw_shape(a.W, a.C, a.Co) == 2 ? 96 : 32))) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return aligned16(a.in) && aligned16(a.out) && aligned16(a.w_mwalk) && !(a.in_fs & 3) && !(a.out_fs & 3);
}

const char* mwalk_kernel_label(const BlockArgs& a, char* buf, size_t cap) {
    snprintf(buf, cap, "mwalk_kernel<%d,%d,%d>", a.C / 4, a.Co / 16, (a.W + 15) / 16);
    return buf;
}

int launch_mwalk(const BlockArgs& a, void* stream) {
    if (!mwalk_kernel_supports(a)) return (int)hipErrorInvalidValue;
    return mw_shape(a.W, a.C, a.Co) == 2 ? mw_launch<MW24>(a, (hipStream_t)stream) : mw_launch<MW48>(a, (hipStream_t)stream);
}

}  // namespace mi
