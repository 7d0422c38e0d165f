// tail_kernels.hip — stage programs for the small-spatial tails, SEVERAL FRAMES PER WORKGROUP, on v_mfma_f32_16x16x4_f32.
//
// The frame-resident stage programs of resident_kernels.hip give one frame to a 512-thread workgroup and cut every stage
// into units of 32 pixels x 32 channels on v_mfma_f32_32x32x2_f32, the weights of a unit streamed from L2 by the wave that
// runs it.  From 12x12 down that wastes most of the machine: a 4x4 frame fills half of one 32-pixel tile (four of the eight
// waves idle, every one of them streaming its 16 KB of weights for 16 pixels), a 2x2 frame an eighth; a stage costs its
// ~10 k cycles of fixed latency whatever it computes.  Here
//
//   * a workgroup owns G consecutive frames (G chosen per launch from the batch size and the LDS the program needs); every
//     activation tensor of the program lives in LDS as ONE dense array of G*H*W pixels ([pixel][C + 4] floats, no zero
//     borders: the depthwise stage tests its nine taps against the frame instead and reads a dedicated zero pixel), so the
//     pixels of the G frames fill the 16-pixel MFMA tiles together and the weights of a stage are fetched once per G frames;
//   * a WAVE owns one 16-channel output tile of the stage (8 waves = 128 output channels; narrower stages split the pixel
//     tiles over the waves that would idle) and keeps its A operands (the tile's 16 x K weights: K/4 registers) in registers
//     for all pixel tiles — the weights travel L2 -> registers once per wave and stage;
//   * the B operand (lane = (k-quarter kq, pixel n): K/4 consecutive channels of pixel n, contraction index permuted on the
//     host to match) is read as 16-byte pieces straight from the LDS tensor (1x1 and 2x2-stride-2 convolutions: the four taps
//     of a 2x2 window are the four k-quarters) or from the depthwise scratch;
//   * a depthwise 3x3 (stride 1 or 2, TF SAME) is computed ONCE per pixel and channel by all 512 threads (thread = (pixel,
//     channel quad), its nine taps + bias in registers for the whole stage) into an LDS scratch that the waves then contract;
//     the stride-2 block in front of a chain whose input does not fit in LDS reads its taps from global memory here
//     (16 consecutive lanes = the 256 bytes of a pixel);
//   * D gives lane (kq, n) four consecutive output channels of pixel n: bias + skip (direct or 2x2 max-pool, zero channel
//     pad; from LDS or global memory) + PReLU / ReLU as one float4, stored to the LDS tensor and / or global memory.  A
//     BlazeBlock updates its input tensor in place (its depthwise taps were consumed into the scratch before the barrier;
//     the skip value is read by the lane that overwrites it).
//
// Replaces the same TFLite op chains as resident_kernels.hip / chain_kernels.hip (`interpreter.invoke()` at
// /root/reference/src/face_detection_lite/face_landmark.rs:265, iris_landmark.rs:203).  Exact f32 (the f32 MFMA accumulates
// like an fmaf chain); results differ from the other kernels by the order of the contraction only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "kernels.hpp"
#include "launch.hpp"

namespace mi {

namespace {

typedef float tf32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 tld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void tst4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 tmax4(float4 a, float4 b, float4 c, float4 d) {
    return make_float4(fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y)),
                       fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)), fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w)));
}
__device__ __forceinline__ float* tresolve(const ResBases& bs, const ResRef& r, int frame) {
    return bs.p[r.base] + r.root_off * bs.scale[r.base] + r.inner + (long)(frame + bs.frame0[r.base]) * r.fs;
}
// n / d for n * d < 2^32 with m = ceil(2^32 / d) (host: tail_magic; m = 0 stands for d = 1)
__device__ __forceinline__ int tdiv(int n, unsigned m) { return m ? (int)__umulhi((unsigned)n, m) : n; }

// What the waves need to know about a stage, read from the descriptor once (scalar registers).
struct TailView {
    int P, HW, Ho, Wo, Co;    // P = pixels of this workgroup's frames in the stage's output
    unsigned mHW, mW;
    int n4;                   // float4 steps of a lane's k-quarter (Kv / 16)
    // B operand
    const float* bsrc;        // LDS: tensor / scratch base
    int bPS;                  // pixel stride (floats)
    int gather2;              // 2x2 stride-2 window: k-quarter = tap
    int sH, sW;               // source frame size (gather2)
    // epilogue
    float* dl;                // LDS destination (null: none), pixel stride dPS
    int dPS;
    float* dg;                // global destination of the first frame (null: none), frame stride dfs
    long dfs;
    int res_mode, res_C, rPS, rW;
    const float* rl;          // skip source in LDS (null: global / none)
    const float* rg;
    long rfs;
    float hi;
    const float* wa;          // A operands [tile][n4][lane][4]
    const float* wc;          // [bias 16 nct][slope 16 nct]
};

// The constants a wave needs from global memory for a stage — the first half of its A operands, bias, slope — are fetched while the
// stage BEFORE it computes (PRE variant: a stage is a few thousand cycles, an L2 round trip in front of its MFMAs a quarter of it).
constexpr int kTailPreA = 4;
struct TailPre {
    float4 A[kTailPreA];
    float4 bias, slope;
};
__device__ __forceinline__ void tail_prefetch(const TailStage& st, const float* weights, TailPre& pf) {
    if (st.kind == TAIL_LOAD) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4;
    const int nct = (st.Co + 15) >> 4, n4 = st.Kv >> 4;
    const int ct = nct >= 8 ? wave : wave % nct;
    const float* wa = weights + st.w_a + ((long)ct * n4 * 64 + lane) * 4;
#pragma unroll
    for (int i = 0; i < kTailPreA; i++) pf.A[i] = tld4(wa + 256 * i);   // (a tile of fewer steps: the loads run into the next tile / constant of the blob, which ends with slack)
    const float* wc = weights + st.w_c + 16 * ct + 4 * kq;
    pf.bias = tld4(wc);
    pf.slope = tld4(wc + 16 * nct);
}

#ifdef MI_TAIL_STAMPS
unsigned long long* g_tail_stamps = nullptr;  // [workgroup][stage][4]: stage start, depthwise phase done, units done, closing barrier passed (s_memtime)
#define MI_TAIL_STAMP(k) if (stamps && threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); stamps[((long)blockIdx.x * nstages + s) * 4 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define MI_TAIL_STAMP(k)
#endif

// One output-channel tile x NPT consecutive pixel tiles from tb on (a tile >= t1 repeats tile tb and is not stored).  a0: A holds the first
// k-block of the tile's A operands (n4b <= 8 float4 steps; when nkb == 1 that is all of it and A is left untouched); nkb k-blocks (the
// 2x2 stride-2 convolutions contract over 256 / 512 values: the further blocks are streamed through A).  N4 = float4 steps per k-block
// (compile time: with run-time predicates around the steps the compiler waited for every LDS read right behind its issue).  The stage
// programs run every instruction a few times per stage: code size counts (a version with three tile-group sizes and the constants of the
// depthwise stage prefetched too was 120 KB of code and ran three times slower than this one).
template <int N4, int NPT>
__device__ __forceinline__ void tail_group(const TailView& v, float4 (&A)[8], bool a0, const float* wa, const float4& bias, const float4& slope, int ch, int tb, int t1,
                                           int nkb) {
    const int lane = threadIdx.x & 63, kq = lane >> 4, n = lane & 15;
    const int K4 = 4 * v.n4;
    const float* bp[NPT];
#pragma unroll
    for (int u = 0; u < NPT; u++) {
        const int pc = min((tb + u < t1 ? tb + u : tb) * 16 + n, v.P - 1);
        if (v.gather2) {
            const int f = tdiv(pc, v.mHW), q = pc - f * v.HW, oy = tdiv(q, v.mW), ox = q - oy * v.Wo;
            bp[u] = v.bsrc + (long)((f * v.sH + 2 * oy + (kq >> 1)) * v.sW + 2 * ox + (kq & 1)) * v.bPS;
        } else {
            bp[u] = v.bsrc + (long)pc * v.bPS + kq * K4;
        }
    }
    // every output column accumulates its K products in the same order whatever tile slot it sits in: a frame's result does not
    // depend on its position in the batch or on the frames per workgroup (bit-identical)
    tf32x4 D[NPT];
#pragma unroll
    for (int u = 0; u < NPT; u++) D[u] = tf32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < nkb; kb++) {
        if (kb > 0 || !a0) {
#pragma unroll
            for (int i = 0; i < N4; i++) A[i] = tld4(wa + 256 * (kb * N4 + i));
        }
        float4 b[NPT][2];   // the B operands of step i + 1 are on their way while step i's MFMAs run
#pragma unroll
        for (int u = 0; u < NPT; u++) b[u][0] = tld4(bp[u] + 4 * N4 * kb);
#pragma unroll
        for (int i = 0; i < N4; i++) {
            if (i + 1 < N4) {
#pragma unroll
                for (int u = 0; u < NPT; u++) b[u][(i + 1) & 1] = tld4(bp[u] + 4 * (N4 * kb + i + 1));
            }
#pragma unroll
            for (int u = 0; u < NPT; u++) D[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i].x, b[u][i & 1].x, D[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NPT; u++) D[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i].y, b[u][i & 1].y, D[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NPT; u++) D[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i].z, b[u][i & 1].z, D[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NPT; u++) D[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i].w, b[u][i & 1].w, D[u], 0, 0, 0);
        }
    }
    // ---- epilogue: lane (kq, n) holds output channels ch .. ch + 3 of pixel (tb + u) * 16 + n
#pragma unroll
    for (int u = 0; u < NPT; u++) {
        const int pu = (tb + u) * 16 + n;
        if (tb + u >= t1 || pu >= v.P || ch >= v.Co) continue;
        const int f = tdiv(pu, v.mHW), q = pu - f * v.HW;
        float4 r = make_float4(D[u][0] + bias.x, D[u][1] + bias.y, D[u][2] + bias.z, D[u][3] + bias.w);
        if (v.res_mode != RES_NONE && ch < v.res_C) {
            float4 sk;
            if (v.rl) {
                if (v.res_mode == RES_DIRECT) {
                    sk = tld4(v.rl + (long)pu * v.rPS + ch);
                } else {  // 2x2 max-pool of a (2 Ho x 2 Wo) source
                    const int oy = tdiv(q, v.mW), ox = q - oy * v.Wo;
                    const float* s0 = v.rl + (long)((f * 2 * v.Ho + 2 * oy) * v.rW + 2 * ox) * v.rPS + ch;
                    sk = tmax4(tld4(s0), tld4(s0 + v.rPS), tld4(s0 + (long)v.rW * v.rPS), tld4(s0 + (long)(v.rW + 1) * v.rPS));
                }
            } else if (v.res_mode == RES_DIRECT) {
                sk = tld4(v.rg + f * v.rfs + (long)q * v.res_C + ch);
            } else {
                const int oy = tdiv(q, v.mW), ox = q - oy * v.Wo;
                const float* s0 = v.rg + f * v.rfs + ((long)(2 * oy) * v.rW + 2 * ox) * v.res_C + ch;
                sk = tmax4(tld4(s0), tld4(s0 + v.res_C), tld4(s0 + (long)v.rW * v.res_C), tld4(s0 + (long)(v.rW + 1) * v.res_C));
            }
            r.x += sk.x; r.y += sk.y; r.z += sk.z; r.w += sk.w;
        }
        r.x = fminf(fmaxf(r.x, 0.f) + slope.x * fminf(r.x, 0.f), v.hi);
        r.y = fminf(fmaxf(r.y, 0.f) + slope.y * fminf(r.y, 0.f), v.hi);
        r.z = fminf(fmaxf(r.z, 0.f) + slope.z * fminf(r.z, 0.f), v.hi);
        r.w = fminf(fmaxf(r.w, 0.f) + slope.w * fminf(r.w, 0.f), v.hi);
        if (v.dl) tst4(v.dl + (long)pu * v.dPS + ch, r);
        if (v.dg) {
            float* o = v.dg + f * v.dfs + (long)q * v.Co + ch;
            if ((v.Co & 3) == 0) {
                tst4(o, r);
            } else {
                o[0] = r.x;
                if (ch + 1 < v.Co) o[1] = r.y;
                if (ch + 2 < v.Co) o[2] = r.z;
                if (ch + 3 < v.Co) o[3] = r.w;
            }
        }
    }
}

// One output-channel tile x the pixel tiles [t0, t1) of the stage on this wave, in pairs (two independent MFMA chains share every A
// register), a last odd tile alone.
template <int N4>
__device__ __forceinline__ void tail_tile_n(const TailView& v, float4 (&A)[8], const float4& bias, const float4& slope, int ct, int t0, int t1, int nkb) {
    const int lane = threadIdx.x & 63, kq = lane >> 4;
    const int ch = 16 * ct + 4 * kq;
    const float* wa = v.wa + ((long)ct * v.n4 * 64 + lane) * 4;
    int tb = t0;
    bool a0 = true;
    for (; tb + 1 < t1; tb += 2) { tail_group<N4, 2>(v, A, a0, wa, bias, slope, ch, tb, t1, nkb); a0 = nkb == 1; }
    if (tb < t1) tail_group<N4, 1>(v, A, a0, wa, bias, slope, ch, tb, t1, nkb);
}
__device__ __forceinline__ void tail_tile(const TailView& v, float4 (&A)[8], const float4& bias, const float4& slope, int ct, int t0, int t1, int n4b, int nkb) {
    switch (n4b) {
        case 1: tail_tile_n<1>(v, A, bias, slope, ct, t0, t1, nkb); break;
        case 2: tail_tile_n<2>(v, A, bias, slope, ct, t0, t1, nkb); break;
        case 4: tail_tile_n<4>(v, A, bias, slope, ct, t0, t1, nkb); break;
        default: tail_tile_n<8>(v, A, bias, slope, ct, t0, t1, nkb); break;
    }
}

// PRE: the variant that fetches a stage's constants a stage ahead (256 registers, one workgroup per CU: small batches and programs whose
// LDS leaves no room for a second workgroup anyway); without it the kernel keeps to 128 registers and two workgroups share a CU.
template <bool PRE>
__global__ __launch_bounds__(512, PRE ? 2 : 4) void tail_kernel(const TailStage* __restrict__ prog, int nstages, int G, int B, int zero_off, ResBases bs, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
    const int frame0 = blockIdx.x * G, nf = min(G, B - frame0);
    TailPre pf;
    if (PRE) tail_prefetch(prog[0], bs.weights, pf);
    if (tid < 64) tst4(lds + zero_off + 4 * tid, make_float4(0.f, 0.f, 0.f, 0.f));   // the pixel out-of-frame depthwise taps read (<= 256 channels)
    __syncthreads();
    for (int s = 0; s < nstages; s++) {
        const TailStage& st = prog[s];
        const int kind = st.kind;
        const bool more = s + 1 < nstages;
        MI_TAIL_STAMP(0)
        if (kind == TAIL_LOAD) {
            // dense NHWC frames -> the LDS tensor (pixel stride C + 4)
            if (PRE && more) tail_prefetch(prog[s + 1], bs.weights, pf);
            const int C4 = st.src_C >> 2, per = st.src_H * st.src_W * C4, PS = st.src_C + 4;
            const unsigned mper = st.mHW, mC4 = st.mW;
            const float* src = tresolve(bs, st.src_g, frame0);
            const long fs = st.src_g.fs;
            float* dst = lds + (long)st.dst_off * G;
            const int HW = st.src_H * st.src_W;
            for (int i = tid; i < nf * per; i += 512) {
                const int f = tdiv(i, mper), e = i - f * per, px = tdiv(e, mC4), c4 = e - px * C4;
                tst4(dst + (long)(f * HW + px) * PS + 4 * c4, tld4(src + f * fs + 4 * (long)e));
            }
            MI_TAIL_STAMP(1)
            MI_TAIL_STAMP(2)
            __syncthreads();
            MI_TAIL_STAMP(3)
            continue;
        }
        const int Ho = st.Ho, Wo = st.Wo, HW = Ho * Wo, P = nf * HW;
        const int Kv = st.Kv;
        const int nct = (st.Co + 15) >> 4, n4 = Kv >> 4, n4b = min(n4, 8), nkb = (n4 + 7) >> 3;
        const int ct0 = nct >= 8 ? wave : wave % nct;
        const float* wa0 = bs.weights + st.w_a + ((long)ct0 * n4 * 64 + lane) * 4;
        const float* wc0 = bs.weights + st.w_c + 16 * ct0 + 4 * kq;
        TailView v;
        v.P = P; v.HW = HW; v.Ho = Ho; v.Wo = Wo; v.Co = st.Co; v.mHW = st.mHW; v.mW = st.mW;
        v.n4 = n4;
        v.gather2 = 0; v.sH = st.src_H; v.sW = st.src_W;
        float4 A[8], bias, slope;
        if (kind == TAIL_DW) {
            // ---- depthwise 3x3 (+ bias) of every pixel and channel quad, once: thread = (pixel, quad), taps in registers
            const int KQ = Kv >> 2, kqs = __builtin_ctz(KQ);      // KQ is a power of two <= 64 (planner)
            const int c4 = tid & (KQ - 1), prow = tid >> kqs, PP = 512 >> kqs;
            const float* cw = bs.weights + st.w_c + 32 * nct + 4 * c4;    // [9][Kv] taps, [Kv] bias
            float4 w[9];
#pragma unroll
            for (int t = 0; t < 9; t++) w[t] = tld4(cw + t * Kv);
            const float4 bdw = tld4(cw + 9 * Kv);
            const int PSk = Kv + 4, S = st.S, sH = st.src_H, sW = st.src_W, sC = st.src_C;
            float* scr = lds + (long)st.scr_off * G + 4 * c4;
            const bool from_lds = st.src_off >= 0;
            const int sPS = sC + 4;
            const float* sl = lds + (long)(from_lds ? st.src_off : 0) * G + 4 * c4;
            const float* sg = from_lds ? nullptr : tresolve(bs, st.src_g, frame0) + 4 * c4;
            const long sfs = st.src_g.fs;
            const float* zl = lds + zero_off + 4 * c4;
            float* pool = st.pool_off >= 0 ? lds + (long)st.pool_off * G + 4 * c4 : nullptr;   // 2x2 max-pool of the source = taps (0..1, 0..1): the skip, for the epilogue
            if constexpr (PRE) {
            // an item = two horizontally adjacent output pixels: they share two (stride 1) / one (stride 2) of their three tap columns, so a
            // pair costs 12 / 15 reads instead of 18, and the reads of a pair are in flight together
            const int Wp = (Wo + 1) >> 1, per = Ho * Wp, nitems = nf * per;
            for (int it = prow; it < nitems; it += PP) {
                const int f = tdiv(it, st.mHWp), r = it - f * per, oy = tdiv(r, st.mWp), ox = 2 * (r - oy * Wp);
                const int iy0 = oy * S - st.pt, ix0 = ox * S - st.pl;
                const int b0 = (f * sH + iy0) * sW + ix0;   // source pixel of tap (0, 0) of the first pixel, counted over the workgroup's frames (may lie outside: unused then)
                const int p0 = f * HW + oy * Wo + ox;
                const bool two = ox + 1 < Wo;
                bool rok[3], cok[5];
#pragma unroll
                for (int k = 0; k < 3; k++) rok[k] = (unsigned)(iy0 + k) < (unsigned)sH;
#pragma unroll
                for (int k = 0; k < 5; k++) cok[k] = (unsigned)(ix0 + k) < (unsigned)sW;
                float4 d[3][5];
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 5; kx++) {
                        if (kx == 4 && S == 1) continue;   // (uniform)
                        const bool ok = rok[ky] && cok[kx];
                        if (from_lds) {
                            d[ky][kx] = tld4(ok ? sl + (long)(b0 + ky * sW + kx) * sPS : zl);
                        } else {
                            const float4 x = tld4(sg + f * sfs + (long)(ok ? (iy0 + ky) * sW + ix0 + kx : 0) * sC);
                            d[ky][kx] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                float4 acc0 = bdw, acc1 = bdw;
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {   // (the taps of a pixel in the order of the single-pixel form: ky, kx ascending)
                        const float4 wt = w[ky * 3 + kx], x0 = d[ky][kx];
                        acc0.x = fmaf(x0.x, wt.x, acc0.x); acc0.y = fmaf(x0.y, wt.y, acc0.y); acc0.z = fmaf(x0.z, wt.z, acc0.z); acc0.w = fmaf(x0.w, wt.w, acc0.w);
                    }
                if (S == 1) {
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            const float4 wt = w[ky * 3 + kx], x1 = d[ky][kx + 1];
                            acc1.x = fmaf(x1.x, wt.x, acc1.x); acc1.y = fmaf(x1.y, wt.y, acc1.y); acc1.z = fmaf(x1.z, wt.z, acc1.z); acc1.w = fmaf(x1.w, wt.w, acc1.w);
                        }
                } else {
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            const float4 wt = w[ky * 3 + kx], x1 = d[ky][kx + 2];
                            acc1.x = fmaf(x1.x, wt.x, acc1.x); acc1.y = fmaf(x1.y, wt.y, acc1.y); acc1.z = fmaf(x1.z, wt.z, acc1.z); acc1.w = fmaf(x1.w, wt.w, acc1.w);
                        }
                }
                tst4(scr + (long)p0 * PSk, acc0);
                if (two) tst4(scr + (long)(p0 + 1) * PSk, acc1);
                if (pool) {   // (stride 2, pads 0: the 2x2 windows are columns 0..1 and 2..3 of rows 0..1)
                    tst4(pool + (long)p0 * PSk, tmax4(d[0][0], d[0][1], d[1][0], d[1][1]));
                    if (two) tst4(pool + (long)(p0 + 1) * PSk, tmax4(d[0][2], d[0][3], d[1][2], d[1][3]));
                }
            }
            } else {   // (128 registers: one pixel per item)
            for (int p = prow; p < P; p += PP) {
                const int f = tdiv(p, st.mHW), q = p - f * HW, oy = tdiv(q, st.mW), ox = q - oy * Wo;
                const int iy0 = oy * S - st.pt, ix0 = ox * S - st.pl;
                const int b0 = (f * sH + iy0) * sW + ix0;   // source pixel of tap (0, 0), counted over the workgroup's frames (may lie outside: unused then)
                bool rok[3], cok[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { rok[k] = (unsigned)(iy0 + k) < (unsigned)sH; cok[k] = (unsigned)(ix0 + k) < (unsigned)sW; }
                float4 d[9];
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        const bool ok = rok[ky] && cok[kx];
                        if (from_lds) {
                            d[ky * 3 + kx] = tld4(ok ? sl + (long)(b0 + ky * sW + kx) * sPS : zl);
                        } else {
                            const float4 x = tld4(sg + f * sfs + (long)(ok ? (iy0 + ky) * sW + ix0 + kx : 0) * sC);
                            d[ky * 3 + kx] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                float4 acc = bdw;
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    acc.x = fmaf(d[t].x, w[t].x, acc.x);
                    acc.y = fmaf(d[t].y, w[t].y, acc.y);
                    acc.z = fmaf(d[t].z, w[t].z, acc.z);
                    acc.w = fmaf(d[t].w, w[t].w, acc.w);
                }
                tst4(scr + (long)p * PSk, acc);
                if (pool) tst4(pool + (long)p * PSk, tmax4(d[0], d[1], d[3], d[4]));
            }
            }
            __syncthreads();
            v.bsrc = lds + (long)st.scr_off * G; v.bPS = PSk;
        } else {  // TAIL_GATHER: 1x1, or 2x2 stride 2 (the four taps = the four k-quarters), source in LDS
            v.bsrc = lds + (long)st.src_off * G; v.bPS = st.src_C + 4;
            v.gather2 = st.K == 2;
        }
        MI_TAIL_STAMP(1)
        if (!PRE) {   // (two workgroups per CU: the other one computes while these travel)
#pragma unroll
            for (int i = 0; i < 8; i++) A[i] = tld4(wa0 + 256 * i);
            bias = tld4(wc0); slope = tld4(wc0 + 16 * nct);
        } else {
            // this stage's constants leave the prefetch registers (the second half of A is asked for now and arrives behind the first
            // half's MFMAs), the next stage's start their trip
#pragma unroll
            for (int i = 0; i < kTailPreA; i++) A[i] = pf.A[i];
            if (n4b > kTailPreA) {
#pragma unroll
                for (int i = kTailPreA; i < 8; i++) A[i] = tld4(wa0 + 256 * i);
            }
            bias = pf.bias; slope = pf.slope;
            if (more) tail_prefetch(prog[s + 1], bs.weights, pf);
        }
        v.dl = st.dst_off >= 0 ? lds + (long)st.dst_off * G : nullptr; v.dPS = st.Co + 4;
        v.dg = st.dst_g.base >= 0 ? tresolve(bs, st.dst_g, frame0) : nullptr; v.dfs = st.dst_g.fs;
        v.res_mode = st.res_mode; v.res_C = st.res_C; v.rPS = st.res_C + 4; v.rW = st.res_W;
        v.rl = (st.res_mode != RES_NONE && st.res_off >= 0) ? lds + (long)st.res_off * G : nullptr;
        v.rg = (st.res_mode != RES_NONE && st.res_off < 0) ? tresolve(bs, st.res_g, frame0) : nullptr; v.rfs = st.res_g.fs;
        if (kind == TAIL_DW && st.pool_off >= 0) { v.res_mode = RES_DIRECT; v.rl = lds + (long)st.pool_off * G; v.rg = nullptr; }   // the pooled skip, one pixel per output pixel
        v.hi = st.act == ACT_RELU6 ? 6.f : INFINITY;
        v.wa = bs.weights + st.w_a; v.wc = bs.weights + st.w_c;
        // ---- units: wave = output-channel tile; stages of fewer than 8 tiles split the pixel tiles over the remaining waves
        const int ntl = (P + 15) >> 4;
        const int parts = nct >= 8 ? 1 : 8 / nct, part = nct >= 8 ? 0 : wave / nct;
        const int t0 = part * ntl / parts, t1 = (part + 1) * ntl / parts;
        if (part < parts && t0 < t1) {
            tail_tile(v, A, bias, slope, ct0, t0, t1, n4b, nkb);
            for (int ct = ct0 + 8; ct < nct; ct += 8) {   // stages of more than 128 output channels: the further tiles fetch their constants themselves
                const float* wa = v.wa + ((long)ct * n4 * 64 + lane) * 4;
#pragma unroll
                for (int i = 0; i < 8; i++) A[i] = tld4(wa + 256 * i);
                bias = tld4(v.wc + 16 * ct + 4 * kq);
                slope = tld4(v.wc + 16 * nct + 16 * ct + 4 * kq);
                tail_tile(v, A, bias, slope, ct, t0, t1, n4b, nkb);
            }
        }
        MI_TAIL_STAMP(2)
        __syncthreads();
        MI_TAIL_STAMP(3)
    }
}

}  // namespace

unsigned tail_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

// Can the kernel run this stage (the planner asks before it builds a program)?
bool tail_stage_ok(int kind, int K, int S, int src_C, int Kv, int Co, bool dst_lds) {
    if (kind == TAIL_LOAD) return src_C % 4 == 0;
    if (Kv % 16 || Kv > 512 || (Kv > 128 && Kv % 128) || (Kv < 128 && (Kv & (Kv - 1)))) return false;   // float4 steps per k-quarter: 1, 2, 4, 8 per k-block
    if (Co < 1 || Co > 512 || (dst_lds && Co % 4)) return false;
    if (kind == TAIL_DW) {
        if (K != 3 || (S != 1 && S != 2) || Kv != src_C || Kv > 256) return false;
        const int KQ = Kv / 4;
        return (KQ & (KQ - 1)) == 0 && KQ <= 64;
    }
    if (kind == TAIL_GATHER) return (K == 1 && S == 1 && Kv == src_C) || (K == 2 && S == 2 && Kv == 4 * src_C);
    return false;
}

int launch_tail(const TailLaunch& a, void* stream) {
    if (!a.prog || a.nstages < 1 || a.B < 1 || a.G < 1 || a.frame_floats < 1) return (int)hipErrorInvalidValue;
    const long zero_off = (long)a.frame_floats * a.G;
    const long lds_bytes = (zero_off + 256) * 4;
    if (lds_bytes > 160 * 1024) return (int)hipErrorInvalidValue;
    unsigned long long* stamps = nullptr;
#ifdef MI_TAIL_STAMPS
    stamps = g_tail_stamps;
    if (const char* only = getenv("MI_TAIL_STAMP_NST"))   // keep the stamps of the launch with this many stages
        if (atoi(only) != a.nstages) stamps = nullptr;
#endif
    const unsigned nwg = (unsigned)((a.B + a.G - 1) / a.G);
    // constants a stage ahead (one workgroup per CU) where a second workgroup would not fit
    // (round 5, second collection: the 256-register form only where a second workgroup would not fit into the CU's LDS anyway.  It used to be
    // chosen for every launch of at most one workgroup per CU as well; measured with the whole pipeline around it the 128-register form is the faster
    // one there — config 5 2.113 -> 2.061 ms with two batches in flight, 2.698 -> 2.638 with one, config 3 0.837 -> 0.828 — MI_TAIL_RULE=0 brings
    // the old rule back)
    static const int rule = getenv("MI_TAIL_RULE") ? atoi(getenv("MI_TAIL_RULE")) : 1;   // tuning aid
    bool pre = a.variant == 1 || (a.variant == 0 && lds_bytes > 80 * 1024);
    if (a.variant == 0 && rule == 0) pre = lds_bytes > 80 * 1024 || nwg <= (unsigned)device_cu_count();
    auto kern = pre ? tail_kernel<true> : tail_kernel<false>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    return (int)launch_kernel(kern, dim3(nwg), dim3(512), (size_t)lds_bytes, (hipStream_t)stream, a.prog, a.nstages, a.G, a.B, (int)zero_off, a.bases, stamps);
}

}  // namespace mi

#ifdef MI_TAIL_STAMPS
// stamps build only (tools/tail_stamps.py): where the per-stage s_memtime stamps of the next launches go
extern "C" void mi_debug_set_tail_stamps(unsigned long long* p) { mi::g_tail_stamps = p; }
#endif
