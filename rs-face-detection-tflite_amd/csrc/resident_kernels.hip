// resident_kernels.hip — frame-resident stage programs (fuse level 5) for the small-spatial parts of the networks.
//
// From 8x8 down (iris), 6x6 down (face mesh), 12x12 down (full-range detector) a layer is a few hundred pixels: one launch
// per fused block is pure latency (dozens of launches of 30-80 us each at batch 1024 for ~1 MFLOP of work per frame), and
// the k x k / stride-k convolutions between the resolutions (2x2 s2, the 3x3 s3 and 2x2 output heads) ran on the generic
// direct-convolution kernel.  Here ONE 512-thread workgroup owns a frame and executes a whole list of stages on it, the
// activations staying in LDS between stages (they reach HBM only where a later launch or the caller needs them):
//
//   GATHER : out = act( W . gather(src) + bias + skip )   k x k convolution with stride k, VALID (1x1 = pointwise); the
//            k*k*C "virtual channels" of an output pixel are its k x k source pixels, read in place (LDS or global)
//   DW     : out = act( W . (DW3x3_s(src) + b_dw) + bias + skip )        the BlazeBlock (stride 1 or 2, TF SAME)
//   LOAD   : dense NHWC frame in global memory -> zero-bordered LDS tensor
//
// (the same TFLite op chains as block_kernels.hip: DEPTHWISE_CONV_2D -> CONV_2D -> ADD -> PRELU/RELU, MAX_POOL_2D / channel
// PAD on the skip path, CONV_2D k x k; `interpreter.invoke()` at /root/reference/src/face_detection_lite/
// iris_landmark.rs:203, face_landmark.rs:265, face_detection.rs:235.)
//
// Mapping: the contraction of a stage is cut into units of 32 output pixels x one 32-channel tile; the 8 waves take units
// round-robin.  A unit is the block kernel's inner loop: lane = (pixel l & 31, k-half l >> 5) reads the B operand of
// v_mfma_f32_32x32x2_f32 for 4 channels of its k-half (one 16-byte gather), the weights arrive from L2 in A-fragment order
// through a register ring that runs 8 k-chunks ahead (a stage is a chain of dependent MFMAs fed by L2 latency; the ring
// loop has no control flow inside, so that the compiler's vmcnt waits stay partial), D gives each lane 4 consecutive
// output channels per register quad for the float4 epilogue (+ bias + skip [direct | 2x2 max-pool, zero channel-padded]
// -> activation -> LDS and/or global).  A DW stage first runs the depthwise 3x3 for a batch of pixel groups with all 512
// threads (each result computed once) into an LDS scratch, which the units then contract like a pointwise source.
// The small per-stage constants (depthwise taps, biases, slopes) of stage s+1 are fetched into registers while stage s
// computes and land in the other half of a double-buffered LDS area behind the stage's closing barrier.
// Exact f32 (MFMA f32 = fmaf chain); results match block_kernels.hip to reassociation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "kernels.hpp"
#include "launch.hpp"

namespace mi {

namespace {

typedef float rf32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 rld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 rmax4(float4 a, float4 b, float4 c, float4 d) {
    return make_float4(fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y)),
                       fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)), fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w)));
}
// LDS traffic between the lanes of ONE wave: the hardware executes a wave's LDS instructions in order; this only stops the
// compiler from moving accesses across the hand-over.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float* resolve(const ResBases& bs, const ResRef& r, int frame) {
    return bs.p[r.base] + r.root_off * bs.scale[r.base] + r.inner + (long)(frame + bs.frame0[r.base]) * r.fs;
}

#ifdef MI_RES_STAMPS
unsigned long long* g_res_stamps = nullptr;  // [workgroup][stage][8]: stage start, constants issued, units done, stage end; wave 0's first unit: start, ring issued, MFMA loop done, epilogue done (s_memtime)
#define MI_RES_STAMP(k) if (stamps && tid == 0) stamps[((long)blockIdx.x * nstages + s) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#define MI_RES_USTAMP(k) if (ustamps && threadIdx.x == 0 && u == 0) { __builtin_amdgcn_sched_barrier(0); ustamps[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define MI_RES_STAMP(k)
#define MI_RES_USTAMP(k)
#endif

// q / d for a wave-uniform divisor: the layer widths and channel-quad counts here are mostly powers of two (a shift instead
// of the ~40-instruction software division; the stage programs are VALU-issue-bound, every instruction counts)
struct Div {
    int d, sh;  // sh >= 0: d == 1 << sh
    __device__ __forceinline__ explicit Div(int dd) : d(dd), sh((dd & (dd - 1)) == 0 ? __builtin_ctz(dd) : -1) {}
    __device__ __forceinline__ int div(int q) const { return sh >= 0 ? q >> sh : q / d; }
};

constexpr int kPF = 8;       // k-chunks of A fragments in flight per lane (LDS sources)
constexpr int kPFG = 4;      // ... when the B fragments come from global memory too (they ride the same ring)
constexpr int kNPre = (kResConstMax / 4 + 511) / 512;  // float4s per thread

// Where a unit reads its B operand.  linear: the depthwise scratch, indexed by (pixel - px_base); otherwise a tensor
// (LDS: off/row/PS/b; global: base pointer, row = W, PS = C, b = 0) gathered over KW x KW taps with stride S.
struct SrcView {
    const float* g;   // global base (MODE 1)
    int off, row, PS, b, C, KW, S, Kv, px_base;
    int slab;   // MODE 3: LDS floats of the 8 waves' staging slabs
    bool linear;
};

// Everything the epilogue needs, read from the descriptor ONCE per stage: the descriptor lives in global memory, and a
// field read after a store has to be loaded again (the compiler cannot rule out aliasing) — scalar-load round trips in the
// middle of the epilogue were a third of a stage's time.
struct EpiView {
    int Ho, Wo, Co, MT;
    int dst_off, dst_row, dst_PS, dst_b;
    bool dst_lds;   // the output goes to LDS (dst_off itself may be shifted below zero by a row band's halo row)
    float* dstg;
    int res_mode, res_C, res_off, res_row, res_PS, res_b, res_W;
    const float* resg;
    float hi;
    const float* wpw;
};
__device__ __forceinline__ EpiView make_epi(const ResStage& st, const ResBases& bs, int frame) {
    EpiView e;
    e.Ho = st.Ho; e.Wo = st.Wo; e.Co = st.Co; e.MT = (st.Co + 31) >> 5;
    e.dst_lds = st.dst_off >= 0; e.dst_off = st.dst_off; e.dst_b = st.dst_b; e.dst_row = st.Wo + 2 * st.dst_b; e.dst_PS = st.dst_PS;
    e.dstg = st.dst_g.base >= 0 ? resolve(bs, st.dst_g, frame) : nullptr;
    e.res_mode = st.res_mode; e.res_C = st.res_C; e.res_off = st.res_off; e.res_b = st.res_b; e.res_W = st.res_W;
    e.res_row = st.res_W + 2 * st.res_b; e.res_PS = st.res_PS;
    e.resg = (st.res_mode != RES_NONE && st.res_off < 0) ? resolve(bs, st.res_g, frame) : nullptr;
    e.hi = st.act == ACT_RELU6 ? 6.f : INFINITY;
    e.wpw = bs.weights + st.w_pw;
    return e;
}

// MODE 0: B from LDS, 1: B from global memory, 2: B from LDS with the lane's virtual channels contiguous (pointwise sources
// and the depthwise scratch: a plain pointer walk, no tap bookkeeping — VALU work does not overlap a wave's own dependent
// MFMAs on this part, so every VALU op in the k-loop is paid in full), 3: pointwise from global memory in blocks of 16
// channels staged through a wave-private LDS slab (the direct gather of mode 1 reads 16 B pieces 128 B apart: with 8 waves
// the touched lines overflow the L1 and every line is fetched from L2 eight times).  Units: pixel groups [pg0, pg1) x all channel tiles.
template <int MODE>
__device__ __forceinline__ void run_units(const EpiView& ep, const SrcView& sv, float* lds, const float* cst_ep, int pg0, int pg1,
                                          unsigned long long* ustamps = nullptr) {
    constexpr int PF = MODE == 1 ? kPFG : kPF;
    constexpr bool CONTIG = MODE == 2 || MODE == 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int npix = ep.Ho * ep.Wo, PGn = pg1 - pg0;
    const int nunits = PGn * ep.MT;
    const int Cp = (sv.Kv + 7) & ~7, Ch = Cp >> 1, nch = Ch >> 2;
    const int Cop = ep.MT * 32;
    const float* biasL = cst_ep;
    const float* alphaL = cst_ep + Cop;
    const int ngroups = nch / PF, rem = nch - ngroups * PF;
    // this lane's first (virtual) channel: tap and channel within the tap (the only divisions, once per stage)
    const int v_first = h * Ch;
    const Div dWo(ep.Wo);
    const int tap0 = v_first / sv.C, c_first = v_first - tap0 * sv.C;
    const int ky0 = tap0 / sv.KW, kx0 = tap0 - ky0 * sv.KW;

    for (int u = wave; u < nunits; u += 8) {
        const int pg = pg0 + u % PGn, mt = u / PGn;
        const int q = pg * 32 + pl;
        const bool valid = q < npix;
        const int qq = valid ? q : 0;
        const int oy = dWo.div(qq), ox = qq - oy * ep.Wo;
        MI_RES_USTAMP(4)
        rf32x16 D;
#pragma unroll
        for (int e = 0; e < 16; e++) D[e] = 0.f;

        const float* wa = ep.wpw + ((long)mt * nch * 64 + lane) * 4;
        // A fragments: a plain pointer walk; past the tile's last chunk it runs into the next tile / the next constant of the
        // weights blob (never used; the blob ends with slack for it)
        auto a_next = [&]() -> float4 {
            const float4 v = rld4(wa);
            wa += 256;
            return v;
        };
        // B operand: 4 consecutive virtual channels of this lane's k-half per chunk, walked tap by tap without divisions
        const long pix_base = sv.linear ? (long)sv.off + (long)(qq - sv.px_base) * sv.PS
                                        : (long)sv.off + ((long)(oy * sv.S + sv.b) * sv.row + ox * sv.S + sv.b) * sv.PS;
        int bv = v_first, bc = c_first, bkx = kx0;
        long btap = ((long)ky0 * sv.row + kx0) * sv.PS;
        const float* bp = lds + pix_base + v_first;  // CONTIG: weights of padded channels are zero and so are the LDS pad lanes
        auto b_next = [&]() -> float4 {
            if (CONTIG) {
                const float4 v = rld4(bp);
                bp += 4;
                return v;
            }
            const bool ok = bv < sv.Kv;
            const long a = pix_base + (ok ? btap + bc : 0);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == 1) { if (ok) v = rld4(sv.g + a); }
            else { v = rld4(lds + a); if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f); }
            bv += 4; bc += 4;
            if (bc >= sv.C) {
                bc = 0; bkx++; btap += sv.PS;
                if (bkx == sv.KW) { bkx = 0; btap += (long)(sv.row - sv.KW) * sv.PS; }
            }
            return v;
        };
        auto mfma4 = [&](const float4& av, const float4& bf) {
            D = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf.x, D, 0, 0, 0);
            D = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf.y, D, 0, 0, 0);
            D = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf.z, D, 0, 0, 0);
            D = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf.w, D, 0, 0, 0);
        };

        if constexpr (MODE == 3) {
            // ---- K-blocked, LDS-staged: block b = channels [16b, 16b+16): lane loads 16 B of pixel (lane>>2) + 16i (a contiguous
            // 64 B per pixel, 1 KiB per instruction), the slab row is [pixel][16 + 4 pad]; chunk 2b + jj reads slab[pl][8h + 4jj].
            constexpr int SP = 20;
            float* slab = lds + sv.slab + wave * (32 * SP);
            const int nblk = sv.Kv >> 4;
            const int lp = lane >> 2, piece = lane & 3;
            const int q0 = min(pg * 32 + lp, npix - 1), q1 = min(pg * 32 + lp + 16, npix - 1);  // clamped: rows past the end are never stored
            const float* g0 = sv.g + (long)q0 * sv.C + 4 * piece;
            const float* g1 = sv.g + (long)q1 * sv.C + 4 * piece;
            float* w0 = slab + lp * SP + 4 * piece;
            const float* rd = slab + pl * SP + 8 * h;
            float4 ringA3[4];
#pragma unroll
            for (int r = 0; r < 4; r++) ringA3[r] = a_next();
            float4 x0 = rld4(g0), x1 = rld4(g1);
            auto do_block = [&](float4& A0, float4& A1, int b) {  // A0 / A1: the ring slots of chunks 2b and 2b + 1 (static registers)
                *reinterpret_cast<float4*>(w0) = x0;
                *reinterpret_cast<float4*>(w0 + 16 * SP) = x1;
                const int bn = min(b + 1, nblk - 1);  // the next block's pixels are on their way while this one is contracted
                x0 = rld4(g0 + 16 * bn);
                x1 = rld4(g1 + 16 * bn);
                wave_sync();
                const float4 b0 = rld4(rd), b1 = rld4(rd + 4);
                wave_sync();  // the slab may be rewritten once every lane has read it (LDS executes a wave's accesses in order)
                mfma4(A0, b0);
                A0 = a_next();
                mfma4(A1, b1);
                A1 = a_next();
            };
            int b = 0;
            for (; b + 1 < nblk; b += 2) {
                do_block(ringA3[0], ringA3[1], b);
                do_block(ringA3[2], ringA3[3], b + 1);
            }
            if (b < nblk) do_block(ringA3[0], ringA3[1], b);
        } else {
        float4 ringA[PF], ringB[MODE == 1 ? PF : 1];
#pragma unroll
        for (int r = 0; r < PF; r++) {
            ringA[r] = a_next();
            if (MODE == 1) ringB[MODE == 1 ? r : 0] = b_next();
        }
        MI_RES_USTAMP(5)
        // LDS sources: the B fragment of chunk j+1 is read before the MFMAs of chunk j (a dependent ds_read per chunk would add
        // its ~130 cycles of latency to every 256 cycles of MFMA); global sources ride the ring
        float4 bcur = MODE == 1 ? make_float4(0.f, 0.f, 0.f, 0.f) : b_next();  // (CONTIG reads one chunk past the pixel at the end: inside the scratch / tensor slack)
        for (int g = 0; g < ngroups; g++) {
#pragma unroll
            for (int r = 0; r < PF; r++) {  // no control flow around the loads: slot r is consumed, then refilled PF chunks ahead
                if (MODE == 1) {
                    mfma4(ringA[r], ringB[MODE == 1 ? r : 0]);
                    ringB[MODE == 1 ? r : 0] = b_next();
                } else {
                    const float4 bn = b_next();
                    mfma4(ringA[r], bcur);
                    bcur = bn;
                }
                ringA[r] = a_next();
            }
        }
#pragma unroll
        for (int r = 0; r < PF - 1; r++)
            if (r < rem) {
                if (MODE == 1) {
                    mfma4(ringA[r], ringB[MODE == 1 ? r : 0]);
                } else {
                    const float4 bn = b_next();
                    mfma4(ringA[r], bcur);
                    bcur = bn;
                }
            }
        }
        MI_RES_USTAMP(6)
        // ---- epilogue: lane holds pixel q, output channels mt*32 + 8*gq + 4*h .. +3 in D[4*gq .. 4*gq+3]
        if (valid) {
            float* dl = ep.dst_lds ? lds + ep.dst_off + ((oy + ep.dst_b) * ep.dst_row + ox + ep.dst_b) * ep.dst_PS : nullptr;
            float* dg = ep.dstg ? ep.dstg + ((long)oy * ep.Wo + ox) * ep.Co : nullptr;
            const float* rl = lds + ep.res_off + (ep.res_mode == RES_DIRECT ? ((oy + ep.res_b) * ep.res_row + ox + ep.res_b) * ep.res_PS
                                                                            : ((2 * oy + ep.res_b) * ep.res_row + 2 * ox + ep.res_b) * ep.res_PS);
            const float* rg = ep.resg ? ep.resg + (ep.res_mode == RES_DIRECT ? ((long)oy * ep.Wo + ox) * ep.res_C : ((long)(2 * oy) * ep.res_W + 2 * ox) * ep.res_C) : nullptr;
            float4 outv[4];
            bool on[4];
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {  // all reads first ...
                const int ch = mt * 32 + 8 * gq + 4 * h;
                on[gq] = ch < ep.Co;
                const int chc = on[gq] ? ch : 0;
                const float4 bb = rld4(biasL + chc), al = rld4(alphaL + chc);
                float4 v = make_float4(D[4 * gq] + bb.x, D[4 * gq + 1] + bb.y, D[4 * gq + 2] + bb.z, D[4 * gq + 3] + bb.w);
                if (ep.res_mode != RES_NONE && chc < ep.res_C) {
                    float4 rv;
                    if (ep.res_off >= 0) {
                        if (ep.res_mode == RES_DIRECT) rv = rld4(rl + chc);
                        else rv = rmax4(rld4(rl + chc), rld4(rl + ep.res_PS + chc), rld4(rl + ep.res_row * ep.res_PS + chc), rld4(rl + (ep.res_row + 1) * ep.res_PS + chc));
                    } else {
                        if (ep.res_mode == RES_DIRECT) rv = rld4(rg + chc);
                        else rv = rmax4(rld4(rg + chc), rld4(rg + ep.res_C + chc), rld4(rg + (long)ep.res_W * ep.res_C + chc), rld4(rg + (long)(ep.res_W + 1) * ep.res_C + chc));
                    }
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
                v.x = fminf(fmaxf(v.x, 0.f) + al.x * fminf(v.x, 0.f), ep.hi);
                v.y = fminf(fmaxf(v.y, 0.f) + al.y * fminf(v.y, 0.f), ep.hi);
                v.z = fminf(fmaxf(v.z, 0.f) + al.z * fminf(v.z, 0.f), ep.hi);
                v.w = fminf(fmaxf(v.w, 0.f) + al.w * fminf(v.w, 0.f), ep.hi);
                outv[gq] = v;
            }
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {  // ... then the stores
                if (!on[gq]) continue;
                const int ch = mt * 32 + 8 * gq + 4 * h;
                const float4 v = outv[gq];
                if (dl) *reinterpret_cast<float4*>(dl + ch) = v;   // Co % 4 == 0 whenever the output stays in LDS (planner)
                if (dg) {
                    if ((ep.Co & 3) == 0) {
                        *reinterpret_cast<float4*>(dg + ch) = v;
                    } else {  // ragged output heads (213, 15, 1 channels)
                        dg[ch] = v.x;
                        if (ch + 1 < ep.Co) dg[ch + 1] = v.y;
                        if (ch + 2 < ep.Co) dg[ch + 2] = v.z;
                        if (ch + 3 < ep.Co) dg[ch + 3] = v.w;
                    }
                }
            }
        }
        MI_RES_USTAMP(7)
    }
}

__device__ __forceinline__ int const_count(const ResStage& st) {  // floats, a multiple of 4
    if (st.kind == RES_STAGE_LOAD) return 0;
    const int Cp = (st.Kv + 7) & ~7, Cop = ((st.Co + 31) >> 5) * 32;
    return (st.kind == RES_STAGE_DW ? 10 * Cp : 0) + 2 * Cop;
}

__global__ __launch_bounds__(512, 4) void resident_kernel(const ResStage* __restrict__ prog, int nstages, int const_off, int const_floats, int bands, ResBases bs,
                                                          unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int frame = blockIdx.x / bands, band = blockIdx.x - frame * bands;
    auto zero_dst = [&](const ResStage& st) {
        const int n4 = st.zero_dst >> 2, off = st.dst_off;
        for (int i = tid; i < n4; i += 512) reinterpret_cast<float4*>(lds + off)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    {   // stage 0: constants straight into the first half, destination cleared
        const ResStage& st = prog[0];
        const int n4 = const_count(st) >> 2;
        const float* cb = bs.weights + (n4 > 0 ? st.cblob : 0);
        for (int i = tid; i < n4; i += 512) reinterpret_cast<float4*>(lds + const_off)[i] = rld4(cb + 4 * (long)i);
        zero_dst(st);
    }
    __syncthreads();
    for (int s = 0; s < nstages; s++) {
        const ResStage& st = prog[s];
        unsigned long long* ust = nullptr;
#ifdef MI_RES_STAMPS
        if (stamps) ust = stamps + ((long)blockIdx.x * nstages + s) * 8;
#endif
        MI_RES_STAMP(0)
        const float* cst = lds + const_off + (s & 1) * const_floats;
        // ---- the next stage's constants: loads issued now, consumed (written to the other half) after this stage's work
        float4 pre[kNPre];
        const bool more = s + 1 < nstages;
        const int npre = more ? const_count(prog[s + 1]) >> 2 : 0;
        const float* cnext = bs.weights + (npre > 0 ? prog[s + 1].cblob : 0);
#pragma unroll
        for (int k = 0; k < kNPre; k++) {
            const int i = tid + 512 * k;
            pre[k] = rld4(cnext + 4 * (long)min(i, max(npre - 1, 0)));
        }
        MI_RES_STAMP(1)
        if (st.kind == RES_STAGE_LOAD) {
            const float* src = resolve(bs, st.src_g, frame);
            const int sW = st.src_W, C4 = st.src_C >> 2, n = st.src_H * sW * C4;
            const int db = st.dst_b, drow = sW + 2 * db, dPS = st.dst_PS, doff = st.dst_off;
            const Div dC4(C4), dW(sW);
            for (int i = tid; i < n; i += 512) {
                const int px = dC4.div(i), c4 = i - px * C4;
                const int y = dW.div(px), x = px - y * sW;
                *reinterpret_cast<float4*>(lds + doff + ((y + db) * drow + x + db) * dPS + 4 * c4) = rld4(src + 4 * (long)i);
            }
        } else if (st.kind == RES_STAGE_DW) {
            EpiView ep = make_epi(st, bs, frame);
            if (st.band_role == 2) {  // this band's rows [r0, r1) of the image
                const int r0 = band * st.band_rows, r1 = min(st.band_H, r0 + st.band_rows);
                ep.Ho = r1 - r0;
                if (ep.dstg) ep.dstg += (long)r0 * st.Wo * st.Co;
                if (ep.resg) ep.resg += (long)r0 * st.Wo * st.res_C;
            }
            const int Kv = st.Kv, Cp = (Kv + 7) & ~7, C4p = Cp >> 2, PSs = Cp + 4;
            const int Wo = st.Wo, npix = ep.Ho * Wo, PGn = (npix + 31) >> 5, dw_pg = st.dw_pg, dw_off = st.dw_off;
            const int srow = st.src_W + 2 * st.src_b, sPS = st.src_PS, S = st.S;
            const int src0 = st.src_off + ((st.src_b - st.pt) * srow + st.src_b - st.pl) * sPS;  // tap (0,0) of output pixel (0,0)
            const Div dC4p(C4p), dWo(Wo);
            SrcView sv;
            sv.g = nullptr; sv.off = dw_off; sv.row = 0; sv.PS = PSs; sv.b = 0; sv.C = Cp; sv.KW = 1; sv.S = 1; sv.Kv = Kv; sv.linear = true;
            for (int pg0 = 0; pg0 < PGn; pg0 += dw_pg) {
                const int pg1 = min(PGn, pg0 + dw_pg);
                const int px0 = pg0 * 32, px1 = min(npix, pg1 * 32);
                // depthwise 3x3 (+ bias) of pixels [px0, px1), every channel quad once: (pixel, quad) items over all threads
                for (int it = tid; it < (px1 - px0) * C4p; it += 512) {
                    const int pxl = dC4p.div(it), c0 = 4 * (it - pxl * C4p);
                    const int q = px0 + pxl, oy = dWo.div(q), ox = q - oy * Wo;
                    const float* p = lds + src0 + (oy * S * srow + ox * S) * sPS + c0;
                    float4 acc = rld4(cst + 9 * Cp + c0);  // weights and the LDS pad lanes are zero above the real channel count
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            const float4 w = rld4(cst + (ky * 3 + kx) * Cp + c0);
                            const float4 d = rld4(p + (ky * srow + kx) * sPS);
                            acc.x = fmaf(d.x, w.x, acc.x);
                            acc.y = fmaf(d.y, w.y, acc.y);
                            acc.z = fmaf(d.z, w.z, acc.z);
                            acc.w = fmaf(d.w, w.w, acc.w);
                        }
                    *reinterpret_cast<float4*>(lds + dw_off + pxl * PSs + c0) = acc;
                }
                __syncthreads();
                sv.px_base = px0;
                run_units<2>(ep, sv, lds, cst + 10 * Cp, pg0, pg1, ust);
                if (pg1 < PGn) __syncthreads();  // the scratch is rewritten by the next batch
            }
        } else {
            EpiView ep = make_epi(st, bs, frame);
            SrcView sv;
            sv.linear = false; sv.px_base = 0; sv.C = st.src_C; sv.KW = st.KW; sv.S = st.S; sv.Kv = st.Kv;
            long src_rows = 0;
            if (st.band_role == 1) {  // rows [r0 - 1, r1 + 1) of the image, clipped: local row 0 = image row lo
                const int r0 = band * st.band_rows, r1 = min(st.band_H, r0 + st.band_rows);
                const int lo = max(0, r0 - 1), hi = min(st.band_H, r1 + 1);
                ep.Ho = hi - lo;
                ep.dst_off += (lo - r0) * ep.dst_row * ep.dst_PS;  // LDS row of image row y is y - r0 + border
                src_rows = lo;
            }
            const int PGn = (ep.Ho * st.Wo + 31) >> 5;
            if (st.src_off >= 0) {
                sv.g = nullptr; sv.off = st.src_off; sv.row = st.src_W + 2 * st.src_b; sv.PS = st.src_PS; sv.b = st.src_b;
                if (sv.KW == 1) run_units<2>(ep, sv, lds, cst, 0, PGn, ust);
                else run_units<0>(ep, sv, lds, cst, 0, PGn, ust);
            } else {
                sv.g = resolve(bs, st.src_g, frame) + src_rows * st.src_W * st.src_C; sv.off = 0; sv.row = st.src_W; sv.PS = st.src_C; sv.b = 0;
                sv.slab = st.dw_off;
                if (st.kblk == 16) run_units<3>(ep, sv, lds, cst, 0, PGn, ust);
                else run_units<1>(ep, sv, lds, cst, 0, PGn, ust);
            }
        }
        MI_RES_STAMP(2)
        if (more) {
            float* nxt = lds + const_off + ((s + 1) & 1) * const_floats;
#pragma unroll
            for (int k = 0; k < kNPre; k++) {
                const int i = tid + 512 * k;
                if (i < npre) reinterpret_cast<float4*>(nxt)[i] = pre[k];
            }
        }
        __syncthreads();
        if (more && prog[s + 1].zero_dst) {  // uniform
            zero_dst(prog[s + 1]);
            __syncthreads();
        }
        MI_RES_STAMP(3)
    }
}

}  // namespace

int resident_const_floats(const ResStage& st) {
    if (st.kind == RES_STAGE_LOAD) return 0;
    const int Cp = (st.Kv + 7) & ~7, Cop = (st.Co + 31) / 32 * 32;
    return (st.kind == RES_STAGE_DW ? 10 * Cp : 0) + 2 * Cop;
}

int launch_resident(const ResLaunch& a, void* stream) {
    if (!a.prog || a.nstages < 1 || a.B < 1 || a.bands < 1 || a.lds_bytes > 160 * 1024 || a.const_floats > kResConstMax) return (int)hipErrorInvalidValue;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(resident_kernel)); e != hipSuccess) return (int)e;
    return (int)launch_kernel(resident_kernel, dim3((unsigned)(a.B * a.bands)), dim3(512), (size_t)a.lds_bytes, (hipStream_t)stream, a.prog, a.nstages,
                              a.const_off, a.const_floats, a.bands, a.bases,
#ifdef MI_RES_STAMPS
                              g_res_stamps
#else
                              nullptr
#endif
    );
}

}  // namespace mi

#ifdef MI_RES_STAMPS
// stamps build only (tools/res_stamps.py): where the per-stage s_memtime stamps of the next launches go
extern "C" void mi_debug_set_res_stamps(unsigned long long* p) { mi::g_res_stamps = p; }
#endif
