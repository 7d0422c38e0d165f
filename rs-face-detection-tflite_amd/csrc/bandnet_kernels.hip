// bandnet_kernels.hip — one launch for a whole BlazeBlock network, for the few frames of a single-image call.
//
// The reference's only operating point is one Mat per call (/root/reference/src/face_detection_lite/face_detection.rs:205-267).
// At one frame the fused per-block launches of the batched plan are 5 - 16 us each, most of it dispatch latency and the start-up
// of a launch that holds a few thousand MACs per lane, and the 16x16x96 chain keeps one CU of 256 busy for 150 us.  Here the
// network behind the first convolution is ONE launch of NW workgroups per frame:
//
//   * a stage is one BlazeBlock (DW3x3 -> PW1x1 -> [+ skip] -> activation), one 1x1 convolution (the SSD heads) or one 2x2 stride-2
//     convolution (the iris network's down-sampling steps, iris_landmark.rs:203: a 1x1 stage whose contraction runs over the four taps,
//     A operands streamed from L2 in rounds of 128 input values);
//   * every stage cuts its output rows into bands; workgroup w owns band w >> wshift of the stages it is active in (the bands get
//     thinner towards the 16x16 and 8x8 stages, where only every 8th / 16th workgroup still has a row), and the owner of output
//     row r owns input row S r: a workgroup's own part of every tensor stays in LDS (two tiles, used in turn);
//   * behind a fork the second branch's stages belong to the workgroups at an offset (woff) inside each group of 1 << wshift, which
//     have had nothing to do since the bands got thinner: both branches run at the same time; the second one's first stage takes all
//     of its input rows from the packets (Rin = 0);
//   * only the rows another workgroup needs — the halo of the 3x3 window — go through global memory, as 16-byte packets
//     {value, tag, value, tag} stored with sc0 sc1 (write-through; the workgroups of a frame sit on all eight XCDs, whose L2s are
//     not coherent with each other).  The reader polls the packets themselves (sc0 sc1 loads) until both tags are the producer
//     stage's: one memory round trip per hand-over, no wait for the stores, no flag.  Measured on MI355X
//     (tools/probes/flag_sync_probe.hip): stores + s_waitcnt + flag + poll + loads cost 2 - 3 us per hand-over, an agent-scope
//     release / acquire pair (L2 write-back + invalidate) 6 - 80 us;
//   * a tag is 64 x generation + stage + 1; the generation is kept in sync[0] and advanced by the last workgroup to finish, and every
//     stage has a packet buffer of its own: nothing has to be cleared between launches and a stale packet never carries the tag a
//     reader waits for;
//   * every wait is bounded: a launch whose workgroups are not all resident (CUs taken by someone else) runs out of iterations,
//     raises *fail and drains; the host repeats that call on the batched plan.  The drain costs ONE limit in all: the workgroup that
//     gives up raises sync[2] (every waiting workgroup looks at it each 64th poll) and a word in its own LDS — a workgroup that has
//     failed, or has seen sync[2], never polls again (its later stages take whatever the first load returned: the results are void).
//
// Per stage and band: [A operands -> registers, small constants -> LDS] [halo rows: packets -> LDS] | DW3x3 on the VALU, one
// thread per (pixel, channel quad) -> LDS | v_mfma_f32_16x16x4_f32 per (16 pixels x 16 output channels) tile, depthwise
// results from LDS, bias / skip / activation in the MFMA result layout -> the other LDS tile, the packet buffer (halo rows), the
// graph output.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>

#include "kernels.hpp"
#include "launch.hpp"

namespace mi {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 512;
constexpr int kSpinLimit = 1 << 12;   // polls of a packet before the launch gives up (a poll is a memory round trip, 1 - 2 us: a few ms — the time the
                                      // batched plan takes for the same call; round 5 waited 2^18 polls = 0.3 s, per stage)
constexpr int kFailCheck = 64;        // ... and every 64th poll looks at sync[2], which the first workgroup to give up raises: the others stop waiting at once
constexpr int kMaxN16 = 8;            // C <= 128
constexpr int kConstFloats = 2048;    // LDS floats of a stage's small constants (32 nct + 10 C <= 1536)

// Packets travel with sc0 sc1 (write-through stores, loads served by memory, not by this XCD's L2), as buffer instructions: hipcc
// counts those itself, so other work may stand between a request and its use (aux: 1 = sc0, 16 = sc1).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kPacketAux = 17;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t packet_buffer(float* base, long floats) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(floats * 4), 0x00020000);
}
__device__ __forceinline__ unsigned poll(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// n / d for 0 <= n < 2^16 through d's magic number m = ceil(2^32 / d) (0 stands for d = 1)
__device__ __forceinline__ int mdiv(int n, unsigned m) { return m ? (int)__umulhi((unsigned)n, m) : n; }

#ifdef MI_BAND_STAMPS
// diagnostic builds only (tools/band_stamps.py): 10 s_memtime stamps per (workgroup of frame 0, stage)
unsigned long long* g_band_stamps = nullptr;
#define MI_BAND_STAMP(k) if (a.stamps && f == 0 && tid == 0) { __builtin_amdgcn_sched_barrier(0); a.stamps[((long)w * 64 + s) * 10 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define MI_BAND_STAMP(k)
#endif

#ifdef MI_BAND_OLDMAP
#define MI_BAND_MYCT(wave, sh) ((wave) >> (sh))
#define MI_BAND_MYPT(wave, sh) ((wave) & ((1 << (sh)) - 1))
#else
#define MI_BAND_MYCT(wave, sh) ((wave) & ((8 >> (sh)) - 1))
#define MI_BAND_MYPT(wave, sh) ((wave) >> (3 - (sh)))
#endif
// A stage descriptor in scalar registers.  The program (BandPacked, 20 dwords a stage) is copied to LDS once — a descriptor read from
// global memory at the top of a stage is a dependent round trip in front of everything else — and a descriptor is then ONE LDS read
// per lane (dword `lane` of it), a v_readlane per dword and scalar bit-field extracts.
struct StageRegs {
    int kind, S, H, W, C, Ho, Wo, Co, R, wshift, nbands, dep, Rin, src_tile, dst_tile, res_tile, pub_lo, pub_hi, src_base, dst_base, res_mode, act, c_floats, res_c;
    int wpc_shift, per_ct;
    unsigned mC4, mWo, mrowq;
    int src_off, dst_off, src_fs, dst_fs, src_ll, dst_ll, w_a, w_c;
    int res_ll, res_stage;   // RES_UP2X: the packet buffer and the stage of the coarse tensor
    int pre;   // a stride-2 BLOCK whose window starts one row / column BEFORE pixel 2r (explicit zero pad in front, VALID behind it: full_range_sparse): S is packed as 3
    int src_lds, dst_lds, res_lds, dst_h3;   // LDS floats in front of the stage's tiles (placed by liveness, sizes of their own); dst_h3: the output tile has R + 3 rows (a stride-2 block reads it), else R + 2
};
// word 0: kind:1 S:2 wshift:4 res_mode:2 act:3 src_tile:2 dst_tile+1:3 pub_lo:1 pub_hi:1 src_base:3 dst_base+1:4 wpc_shift:2 res_tile+1:3 dst_h3:1
// word 1: R:8 Rin:8 dep+1:8 nbands:8    word 2: H:16 W:16    word 3: Ho:16 Wo:16    word 4: C:16 Co:16    word 5: c_floats:12 woff:4 per_ct:16
// words 6-8: mC4 mWo mrowq    words 9-16: src_off dst_off src_fs dst_fs src_ll dst_ll w_a w_c (floats; -1: none)    word 17: res_c:16 (channels of the skip tensor in res_tile) src_lds:12
// word 18: res_ll (RES_UP2X: packet buffer of the coarse tensor, floats)    word 19: res_stage:6 (its producer) dst_lds:12 res_lds:12    (the *_lds: units of 16 floats)
__host__ __device__ inline unsigned bf(unsigned v, int lo, int n) { return (v >> lo) & ((1u << n) - 1u); }
__device__ __forceinline__ int stage_word(const BandPacked* p, int lane) { return reinterpret_cast<const int*>(p)[lane < kBandPackedWords ? lane : 0]; }
__device__ __forceinline__ StageRegs stage_regs(int word) {
    unsigned w[20];
#pragma unroll
    for (int k = 0; k < 20; k++) w[k] = (unsigned)__builtin_amdgcn_readlane(word, k);
    StageRegs r;
    r.kind = bf(w[0], 0, 1); r.S = bf(w[0], 1, 2); r.pre = r.S == 3; r.S = r.pre ? 2 : r.S; r.wshift = bf(w[0], 3, 4); r.res_mode = bf(w[0], 7, 2); r.act = bf(w[0], 9, 3);
    r.src_tile = bf(w[0], 12, 2); r.dst_tile = (int)bf(w[0], 14, 3) - 1; r.pub_lo = bf(w[0], 17, 1); r.pub_hi = bf(w[0], 18, 1);
    r.src_base = bf(w[0], 19, 3); r.dst_base = (int)bf(w[0], 22, 4) - 1; r.wpc_shift = bf(w[0], 26, 2); r.res_tile = (int)bf(w[0], 28, 3) - 1; r.dst_h3 = bf(w[0], 31, 1); r.res_c = bf(w[17], 0, 16); r.src_lds = (int)bf(w[17], 16, 12) << 4;
    r.R = bf(w[1], 0, 8); r.Rin = bf(w[1], 8, 8); r.dep = (int)bf(w[1], 16, 8) - 1; r.nbands = bf(w[1], 24, 8);
    r.H = bf(w[2], 0, 16); r.W = bf(w[2], 16, 16); r.Ho = bf(w[3], 0, 16); r.Wo = bf(w[3], 16, 16); r.C = bf(w[4], 0, 16); r.Co = bf(w[4], 16, 16);
    r.c_floats = bf(w[5], 0, 12); r.per_ct = bf(w[5], 16, 16);
    r.mC4 = w[6]; r.mWo = w[7]; r.mrowq = w[8];
    r.src_off = (int)w[9]; r.dst_off = (int)w[10]; r.src_fs = (int)w[11]; r.dst_fs = (int)w[12];
    r.src_ll = (int)w[13]; r.dst_ll = (int)w[14]; r.w_a = (int)w[15]; r.w_c = (int)w[16];
    r.res_ll = (int)w[18]; r.res_stage = (int)bf(w[19], 0, 6); r.dst_lds = (int)bf(w[19], 6, 12) << 4; r.res_lds = (int)bf(w[19], 18, 12) << 4;
    return r;
}

// CV2: the program has 2x2 stride-2 convolution stages (the iris network).  Their code lives in an instantiation of its own: beside it the other
// stage kinds' code comes out with 37 instead of 9 scalar registers kept in vector lanes and waits on the A registers it reloads, 0.5 us per
// stage on every network (BackCamera 178 -> 204 us; as a second compile-time copy of the stage body inside one kernel the packet loop was
// unswitched into 250 KB of code: the same 0.5 us, from the instruction cache).
// XB: a BLOCK stage of the program takes ALL of its input rows from the packets (the first block of a branch that runs on the workgroups its
// sibling leaves idle: the face mesh) and has to zero its tile's border pixels itself — an instantiation of its own for the same reason (the
// three lines cost every stage of BackCamera 0.04 us).
// WIDE: the program has stages of more than 128 input or output channels (full_range's 12x12 and 6x6 layers) — again an instantiation of its own.
template <bool CV2, bool XB, bool WIDE>
__global__ __launch_bounds__(kThreads) void bandnet_kernel(BandLaunch a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = blockIdx.x, f = blockIdx.y;
    // (no static LDS: the launch may ask for all 160 KB as dynamic LDS.  The generation only changes when the launch has drained.)
    const unsigned base = (unsigned)uni((int)poll(a.sync)) * 64u;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float* const ws = a.base[0] + (long)f * a.ws_frame_floats;
    float* const dwb = lds + a.tiles_floats;   // (behind the tiles, which have places and sizes of their own since round 6)
    float* const lC = dwb + a.dw_floats;
    const BandPacked* const lprog = reinterpret_cast<const BandPacked*>(lC + kConstFloats);
    int* const lfail = reinterpret_cast<int*>(lC + kConstFloats) + a.nstages * (int)(sizeof(BandPacked) / 4);   // this workgroup has given up (16 bytes behind the program)
    {
        const f32x4* gp = reinterpret_cast<const f32x4*>(a.prog);
        f32x4* lp = reinterpret_cast<f32x4*>(lC + kConstFloats);
        for (int i = tid; i < a.nstages * (int)(sizeof(BandPacked) / 16); i += kThreads) lp[i] = gp[i];
        if (tid == 0) *lfail = 0;
    }
    __syncthreads();
    // test hook (engine option "band_test_absent" = k): every k-th workgroup leaves at once and publishes nothing — what its neighbours see when a
    // workgroup of the launch is not resident.  It still counts as finished, so the generation moves on when the others have drained.
    const bool absent = a.absent_mod > 0 && w % a.absent_mod == a.absent_mod - 1;
    // the stages this workgroup takes part in (workgroup w owns band w >> wshift of a stage where its low wshift bits are the stage's woff), as a bit mask:
    // lane k looks at stage k
    unsigned long long amask;
    {
        const int k = lane < a.nstages ? lane : 0;
        const int sh = (int)bf(lprog[k].w[0], 3, 4), nb = (int)bf(lprog[k].w[1], 24, 8), off = (int)bf(lprog[k].w[5], 12, 4);
        amask = __ballot(lane < a.nstages && (w & ((1 << sh) - 1)) == off && (w >> sh) < nb);
    }
    auto next_active = [&](int from) {   // the next such stage after `from` (nstages: none)
        const unsigned long long rest = from + 1 < 64 ? amask >> (from + 1) : 0ull;
        return rest ? from + 1 + (int)__builtin_ctzll(rest) : a.nstages;
    };
    // a stage's A operands for this wave (it keeps one 16-channel output tile for the whole stage) and its small constants
    auto fetch = [&](const StageRegs& st, f32x4 (&A)[kMaxN16], f32x2& A8, float& A4, f32x4& creg) {
        const bool cv2 = CV2 && st.kind != BAND_BLOCK && st.S == 2;   // (its first round of eight 16-value chunks; C % 32 == 0)
        // (wide stages — more than 128 input channels, C % 16 == 0: the first eight chunks here, the others in rounds like a 2x2 convolution's)
        const int nct = (st.Co + 15) >> 4, n16 = cv2 ? kMaxN16 : min(st.C >> 4, kMaxN16), has8 = cv2 ? 0 : (st.C >> 3) & 1, has4 = cv2 ? 0 : (st.C >> 2) & 1;
        const int myct = MI_BAND_MYCT(wave, st.wpc_shift);
        const float* ga = a.consts + st.w_a + (myct < nct ? myct : 0) * st.per_ct;
#pragma unroll
        for (int j = 0; j < kMaxN16; j++) A[j] = j < n16 ? *reinterpret_cast<const f32x4*>(ga + (j * 64 + lane) * 4) : zero4;
        A8 = has8 ? *reinterpret_cast<const f32x2*>(ga + n16 * 256 + lane * 2) : f32x2{0.f, 0.f};
        A4 = has4 ? ga[n16 * 256 + has8 * 128 + lane] : 0.f;
        creg = tid < (st.c_floats >> 2) ? reinterpret_cast<const f32x4*>(a.consts + st.w_c)[tid] : zero4;   // c_floats <= 4 x 512
    };

    int s = absent ? a.nstages : next_active(-1);
    StageRegs st{};
    f32x4 A[kMaxN16], creg = zero4;
    f32x2 A8 = {0.f, 0.f};
    float A4 = 0.f;
    if (s < a.nstages) {
        st = stage_regs(stage_word(lprog + s, lane));
        fetch(st, A, A8, A4, creg);
    }
    while (s < a.nstages) {
        MI_BAND_STAMP(0)
        const bool blk = st.kind == BAND_BLOCK;
        const bool cv2 = CV2 && !blk && st.S == 2;   // 2x2 stride-2 convolution: output row r reads input rows 2r, 2r + 1 (no row above, none below them)
        const int S = st.S, C = st.C, Cs = C + 4, C4 = C >> 2, Co = st.Co, Wo = st.Wo, W = st.W, TW = W + 2;
        const int bi = w >> st.wshift;
        const int r0 = bi * st.R, nro = min(st.Ho, r0 + st.R) - r0, npx = nro * Wo;
        const int p0 = S * r0;                     // first input row of the band; tile row of input row y: y - p0 + 1
        const int Rin = st.dep >= 0 ? min(st.Rin, st.H - p0) : 0;
        // (the WIDE instantiation only: a stride-2 block with its pad in front reads rows 2r - 1 .. 2r + 1 — one row above the band, none below)
        const int pre = WIDE && blk ? st.pre : 0;
        const int ya = blk && (S == 1 || pre) ? p0 - 1 : p0;
        const int yb = blk ? (S == 1 ? p0 + nro + 1 : p0 + 2 * nro + 1 - pre) : (cv2 ? p0 + 2 * nro : p0 + nro);
        float* const tile = lds + st.src_lds;
        const int nct = (Co + 15) >> 4, n16 = C >> 4, has8 = (C >> 3) & 1, has4 = (C >> 2) & 1;
        const unsigned tag_in = base + (unsigned)st.dep + 1u, tag_out = base + (unsigned)s + 1u;
        // wpc = 8 / nct rounded down to a power of two (nct <= 8) waves share an output-channel tile and take its pixel tiles in turn.  The channel
        // tile is the wave's LOW bits — waves w and w + 4 sit on one SIMD, and of a tile's waves only the first has work where a band is one pixel tile
        const int wpc = 1 << st.wpc_shift, myct = MI_BAND_MYCT(wave, st.wpc_shift), mypt = MI_BAND_MYPT(wave, st.wpc_shift);
        const bool wave_on = myct < nct;
        // Wide stages (the WIDE instantiation only; round 6: full_range's 12x12 and 6x6 layers, 144 .. 384 channels): more than eight chunks of input
        // channels are contracted in rounds of eight (A operands of the later rounds read from L2 like a 2x2 convolution's), more than eight output
        // tiles are taken by the waves in turn (tile ct, ct + 8, ct + 16: their A operands read when their turn comes), and the depthwise taps
        // (9 C + C floats: more than the constants' LDS area) are read from L2 by the depthwise phase
        const bool wide = WIDE && !cv2 && (C > 128 || Co > 128);

        // ---- the rows of the input that this workgroup does not own: above [ya, p0) and below [p0 + Rin, yb)
        const int nA = p0 - ya, nH = nA + (yb - (p0 + Rin));
        const int rowq = W * C4, total = nH * rowq;
        // From the producers' packets: at most four elements (eight packets) per lane, all requested at once and again until every tag is
        // the producer stage's.  While the first answers travel, the scalar work of finding the next stage and its descriptor is done.
        const bool packets = st.dep >= 0 && total > 0;
        int dsto[4], pko[4];
        bool real[4];
        u32x4 pa[4], pb[4];
        __amdgpu_buffer_rsrc_t lsrc = packet_buffer(ws + (packets ? st.src_ll : 0), packets ? 2L * st.H * W * C : 0);
        if (packets) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = tid + u * kThreads, ic = min(i, total - 1);
                const int hr = mdiv(ic, st.mrowq), e = ic - hr * rowq, px = mdiv(e, st.mC4), q = e - px * C4;
                const int y = hr < nA ? ya + hr : p0 + Rin + (hr - nA);
                dsto[u] = i < total ? ((y - p0 + 1) * TW + px + 1) * Cs + 4 * q : -1;
                real[u] = i < total && y >= 0 && y < st.H;
                pko[u] = real[u] ? ((y * W + px) * C4 + q) * 32 : 0;   // bytes
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                pa[u] = __builtin_amdgcn_raw_buffer_load_b128(lsrc, pko[u], 0, kPacketAux);
                pb[u] = __builtin_amdgcn_raw_buffer_load_b128(lsrc, pko[u] + 16, 0, kPacketAux);
            }
        }
        if (packets) {
            int it = *lfail ? kSpinLimit : 0;   // a workgroup that has given up once does not wait again (written before the previous stage's closing barrier)
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int u = 0; u < 4; u++) ok = ok && (!real[u] || (pa[u].y == tag_in && pa[u].w == tag_in && pb[u].y == tag_in && pb[u].w == tag_in));
                if (ok) break;
                if (++it > kSpinLimit || (it % kFailCheck == 0 && poll(a.sync + 2) != 0u)) {
                    *a.fail = 1;
                    *lfail = 1;
                    __hip_atomic_store(a.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    pa[u] = __builtin_amdgcn_raw_buffer_load_b128(lsrc, pko[u], 0, kPacketAux);
                    pb[u] = __builtin_amdgcn_raw_buffer_load_b128(lsrc, pko[u] + 16, 0, kPacketAux);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (dsto[u] >= 0) {
                    const f32x4 v = {__uint_as_float(pa[u].x), __uint_as_float(pa[u].z), __uint_as_float(pb[u].x), __uint_as_float(pb[u].z)};
                    *reinterpret_cast<f32x4*>(tile + dsto[u]) = real[u] ? v : zero4;
                }
            if (XB && st.Rin == 0 && blk)   // the first block of a branch that runs on the workgroups its sibling leaves idle: nobody here wrote the tile's border pixels
                for (int i = tid; i < (yb - ya) * 2 * C4; i += kThreads) {
                    const int rr = i / (2 * C4), e = i - rr * 2 * C4, side = e / C4, q = e - side * C4;
                    *reinterpret_cast<f32x4*>(tile + ((ya + rr - p0 + 1) * TW + (side ? W + 1 : 0)) * Cs + 4 * q) = zero4;
                }
        } else if (st.dep < 0) {
            // the program's input: plain memory, complete before the launch
            const float* src = a.base[st.src_base] + st.src_off + (long)f * st.src_fs;
            for (int i = tid; i < total; i += kThreads) {
                const int hr = mdiv(i, st.mrowq), e = i - hr * rowq, px = mdiv(e, st.mC4), q = e - px * C4;
                const int y = hr < nA ? ya + hr : p0 + Rin + (hr - nA);
                f32x4 v = zero4;
                if (y >= 0 && y < st.H) v = *reinterpret_cast<const f32x4*>(src + ((long)y * W + px) * C + 4 * q);
                *reinterpret_cast<f32x4*>(tile + ((y - p0 + 1) * TW + px + 1) * Cs + 4 * q) = v;
            }
            for (int i = tid; i < (yb - ya) * 2 * C4; i += kThreads) {   // nobody wrote this tile's border pixels
                const int rr = i / (2 * C4), e = i - rr * 2 * C4, side = e / C4, q = e - side * C4;
                *reinterpret_cast<f32x4*>(tile + ((ya + rr - p0 + 1) * TW + (side ? W + 1 : 0)) * Cs + 4 * q) = zero4;
            }
        }
        if (tid < (st.c_floats >> 2)) reinterpret_cast<f32x4*>(lC)[tid] = creg;
        // ---- RES_UP2X (the WIDE instantiation: full_range's lateral convolutions, round 6): out = act(W x + b) + the bilinear x2 up-sampling
        // (TFLite ResizeBilinear, half_pixel_centers) of a tensor of half the size, whose two rows this band's ONE output row reads come from their
        // owners' packets into the (unused: a 1x1 stage) depthwise area: [2][Wo / 2][Co + 4]
        int up_y0 = 0, up_y1 = 0;
        float up_dy = 0.f;
        if (WIDE && st.res_mode == RES_UP2X) {
            const int Hc = st.Ho >> 1, Wc = Wo >> 1, Cq = Co >> 2;
            const float iy = ((float)r0 + 0.5f) * 0.5f - 0.5f;
            up_y0 = max((int)floorf(iy), 0);
            up_y1 = min((int)ceilf(iy), Hc - 1);
            up_dy = iy - (float)up_y0;
            const int per_row = Wc * Cq;   // <= 512: one element (two packets) per thread and row
            const unsigned tag_res = base + (unsigned)st.res_stage + 1u;
            __amdgpu_buffer_rsrc_t rsrc = packet_buffer(ws + st.res_ll, 2L * Hc * Wc * Co);
            const bool mine = tid < per_row;
            const int px = mine ? tid / Cq : 0, q = mine ? tid - px * Cq : 0;
            int off[2];
            u32x4 ua[2], ub[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                off[e] = (((e ? up_y1 : up_y0) * Wc + px) * Cq + q) * 32;
                ua[e] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[e], 0, kPacketAux);
                ub[e] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[e] + 16, 0, kPacketAux);
            }
            int it = *lfail ? kSpinLimit : 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int e = 0; e < 2; e++) ok = ok && (!mine || (ua[e].y == tag_res && ua[e].w == tag_res && ub[e].y == tag_res && ub[e].w == tag_res));
                if (ok) break;
                if (++it > kSpinLimit || (it % kFailCheck == 0 && poll(a.sync + 2) != 0u)) {
                    *a.fail = 1;
                    *lfail = 1;
                    __hip_atomic_store(a.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    ua[e] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[e], 0, kPacketAux);
                    ub[e] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[e] + 16, 0, kPacketAux);
                }
            }
            if (mine) {
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const f32x4 v = {__uint_as_float(ua[e].x), __uint_as_float(ua[e].z), __uint_as_float(ub[e].x), __uint_as_float(ub[e].z)};
                    *reinterpret_cast<f32x4*>(dwb + ((e * Wc + px) * (Co + 4)) + 4 * q) = v;
                }
            }
        }
        MI_BAND_STAMP(1)
        __syncthreads();
        MI_BAND_STAMP(2)
        // ---- depthwise 3x3: one thread per (output pixel, channel quad); taps at tile rows trow + S oy + ky
        const int trow = blk && (S == 1 || pre) ? 0 : 1;
        if (blk) {
            if (WIDE && st.c_floats < 32 * nct + 10 * C) {
                // (wide: the taps did not fit the constants' LDS area — c_floats holds bias and slopes only — and come from L2; a code path of its own:
                // a run-time choice between an LDS and a global pointer would make every tap read a FLAT load)
                const float* wdw = a.consts + st.w_c + 32 * nct;
                const float* bdw = wdw + 9 * C;
                for (int i = tid; i < npx * C4; i += kThreads) {
                    const int px = mdiv(i, st.mC4), q = i - px * C4, oy = mdiv(px, st.mWo), ox = px - oy * Wo;
                    const float* t = tile + ((trow + S * oy) * TW + S * ox + (S - 1 - pre)) * Cs + 4 * q;
                    f32x4 k[9];
#pragma unroll
                    for (int j = 0; j < 9; j++) k[j] = *reinterpret_cast<const f32x4*>(wdw + j * C + 4 * q);
                    f32x4 acc = *reinterpret_cast<const f32x4*>(bdw + 4 * q);
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            const f32x4 d = *reinterpret_cast<const f32x4*>(t + (ky * TW + kx) * Cs);
                            acc.x = fmaf(d.x, k[ky * 3 + kx].x, acc.x);
                            acc.y = fmaf(d.y, k[ky * 3 + kx].y, acc.y);
                            acc.z = fmaf(d.z, k[ky * 3 + kx].z, acc.z);
                            acc.w = fmaf(d.w, k[ky * 3 + kx].w, acc.w);
                        }
                    *reinterpret_cast<f32x4*>(dwb + px * Cs + 4 * q) = acc;
                }
            } else {
            const float* wdw = lC + 32 * nct;
            const float* bdw = wdw + 9 * C;
            for (int i = tid; i < npx * C4; i += kThreads) {
                const int px = mdiv(i, st.mC4), q = i - px * C4, oy = mdiv(px, st.mWo), ox = px - oy * Wo;
                const float* t = tile + ((trow + S * oy) * TW + S * ox + (S - 1 - pre)) * Cs + 4 * q;
                f32x4 acc = *reinterpret_cast<const f32x4*>(bdw + 4 * q);
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        const f32x4 d = *reinterpret_cast<const f32x4*>(t + (ky * TW + kx) * Cs);
                        const f32x4 k = *reinterpret_cast<const f32x4*>(wdw + (ky * 3 + kx) * C + 4 * q);
                        acc.x = fmaf(d.x, k.x, acc.x);
                        acc.y = fmaf(d.y, k.y, acc.y);
                        acc.z = fmaf(d.z, k.z, acc.z);
                        acc.w = fmaf(d.w, k.w, acc.w);
                    }
                *reinterpret_cast<f32x4*>(dwb + px * Cs + 4 * q) = acc;
            }
            }
            __syncthreads();
        }
        MI_BAND_STAMP(3)
        // ---- pointwise: (16 pixels x 16 output channels) tiles; the wave keeps its output-channel tile and walks over the pixel tiles
        {
            const int npt = (npx + 15) >> 4;
            const int n = lane & 15, kq = lane >> 4;
            const int Cso = Co + 4, TWo = Wo + 2;
            float* const dtile = st.dst_tile >= 0 ? lds + st.dst_lds : nullptr;
            float* const gout = st.dst_base >= 0 ? a.base[st.dst_base] + st.dst_off + (long)f * st.dst_fs + (long)r0 * Wo * Co : nullptr;
            float* const llo = st.dst_ll >= 0 ? ws + st.dst_ll : nullptr;
            const float hi = st.act == ACT_RELU6 ? 6.f : INFINITY;
            __amdgpu_buffer_rsrc_t osrc = packet_buffer(ws + (llo ? st.dst_ll : 0), llo ? 2L * st.Ho * Wo * Co : 0);
            if (dtile)   // the output band's border pixels
                for (int i = tid; i < (st.R + 2 + st.dst_h3) * 2 * (Co >> 2); i += kThreads) {
                    const int rr = i / (2 * (Co >> 2)), e = i - rr * 2 * (Co >> 2), side = e / (Co >> 2), q = e - side * (Co >> 2);
                    *reinterpret_cast<f32x4*>(dtile + (rr * TWo + (side ? Wo + 1 : 0)) * Cso + 4 * q) = zero4;
                }
            if (wave_on)
              for (int ct = myct; ct < (WIDE ? nct : myct + 1); ct += kThreads / 64) {   // (one turn unless the stage has more than eight output tiles — wpc == 1 then —: no loop at all outside the WIDE instantiation)
                for (int pt = mypt; pt < npt; pt += wpc) {
                    const int px = 16 * pt + n, pxc = min(px, npx - 1);
                    const int oy = mdiv(pxc, st.mWo), ox = pxc - oy * Wo;
                    const float* bp = blk ? dwb + pxc * Cs : tile + ((1 + oy) * TW + ox + 1) * Cs;
                    // four accumulators, one per k-step of a 16-channel chunk: a chain of dependent MFMAs runs at a quarter of the issue rate
                    f32x4 D = zero4, D1 = zero4, D2 = zero4, D3 = zero4;
                    if (cv2) {
                        // contraction over (tap, channel): 4 C / 128 rounds of eight chunks; round 0's A operands came with the stage's
                        // descriptor (A keeps them for the wave's next pixel tile), the others are read here
                        const float* ga = a.consts + st.w_a + myct * st.per_ct;
                        const float* t00 = tile + ((1 + 2 * oy) * TW + 2 * ox + 1) * Cs + 4 * kq;
                        // (a register set of its own for the later rounds: were they loaded into A, every use of A — the other stage kinds' too —
                        // would stand behind a wait for ALL outstanding memory operations, the previous pixel tile's packet stores among them:
                        // measured 0.6 us per stage on every network)
                        const int nr = C >> 5;
                        f32x4 Ar[kMaxN16];
#pragma unroll
                        for (int j = 0; j < kMaxN16; j++) Ar[j] = A[j];
                        for (int r = 0; r < nr; r++) {
                            if (r > 0) {
#pragma unroll
                                for (int j = 0; j < kMaxN16; j++) Ar[j] = *reinterpret_cast<const f32x4*>(ga + ((r * kMaxN16 + j) * 64 + lane) * 4);
                            }
#pragma unroll
                            for (int j = 0; j < kMaxN16; j++) {
                                const int k0 = (r * kMaxN16 + j) * 16;
                                const int tap = (k0 >= C) + (k0 >= 2 * C) + (k0 >= 3 * C);
                                const f32x4 bv = *reinterpret_cast<const f32x4*>(t00 + ((tap >> 1) * TW + (tap & 1)) * Cs + (k0 - tap * C));
                                D = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[j].x, bv.x, D, 0, 0, 0);
                                D1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[j].y, bv.y, D1, 0, 0, 0);
                                D2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[j].z, bv.z, D2, 0, 0, 0);
                                D3 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[j].w, bv.w, D3, 0, 0, 0);
                            }
                        }
                    } else if (wide) {
                        // rounds of eight 16-channel chunks; the first round of the wave's first tile is in A (it came with the stage's descriptor)
                        const float* ga = a.consts + st.w_a + ct * st.per_ct;
                        const int nr = (n16 + kMaxN16 - 1) >> 3;
                        f32x4 Ar[kMaxN16];
#pragma unroll
                        for (int j = 0; j < kMaxN16; j++) Ar[j] = A[j];
                        for (int r = 0; r < nr; r++) {
                            if (r > 0 || ct != myct) {
#pragma unroll
                                for (int j = 0; j < kMaxN16; j++) Ar[j] = r * kMaxN16 + j < n16 ? *reinterpret_cast<const f32x4*>(ga + ((r * kMaxN16 + j) * 64 + lane) * 4) : zero4;
                            }
#pragma unroll
                            for (int j = 0; j < kMaxN16; j++)
                                if (r * kMaxN16 + j < n16) {
                                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + 16 * (r * kMaxN16 + j) + 4 * kq);
                                    D = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[j].x, bv.x, D, 0, 0, 0);
                                    D1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[j].y, bv.y, D1, 0, 0, 0);
                                    D2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[j].z, bv.z, D2, 0, 0, 0);
                                    D3 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[j].w, bv.w, D3, 0, 0, 0);
                                }
                        }
                        if (has8) {   // (a stage that is wide by its OUTPUT may have any input width: 12x12x36 -> 144)
                            const f32x2 a8 = ct == myct ? A8 : *reinterpret_cast<const f32x2*>(ga + n16 * 256 + lane * 2);
                            const f32x2 bv = *reinterpret_cast<const f32x2*>(bp + 16 * n16 + 2 * kq);
                            D = __builtin_amdgcn_mfma_f32_16x16x4f32(a8.x, bv.x, D, 0, 0, 0);
                            D1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a8.y, bv.y, D1, 0, 0, 0);
                        }
                        if (has4) {
                            const float a4 = ct == myct ? A4 : ga[n16 * 256 + has8 * 128 + lane];
                            D2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4, bp[16 * n16 + 8 * has8 + kq], D2, 0, 0, 0);
                        }
                    } else {
#pragma unroll
                    for (int j = 0; j < kMaxN16; j++)
                        if (j < n16) {
                            const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + 16 * j + 4 * kq);
                            D = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j].x, bv.x, D, 0, 0, 0);
                            D1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j].y, bv.y, D1, 0, 0, 0);
                            D2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j].z, bv.z, D2, 0, 0, 0);
                            D3 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j].w, bv.w, D3, 0, 0, 0);
                        }
                    if (has8) {
                        const f32x2 bv = *reinterpret_cast<const f32x2*>(bp + 16 * n16 + 2 * kq);
                        D = __builtin_amdgcn_mfma_f32_16x16x4f32(A8.x, bv.x, D, 0, 0, 0);
                        D1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A8.y, bv.y, D1, 0, 0, 0);
                    }
                    if (has4) D2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A4, bp[16 * n16 + 8 * has8 + kq], D2, 0, 0, 0);
                    }
                    D = (D + D1) + (D2 + D3);
                    // D[i] = output channel 16 ct + 4 kq + i of pixel px
                    const int c0 = 16 * ct + 4 * kq;
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(lC + c0), sl = *reinterpret_cast<const f32x4*>(lC + 16 * nct + c0);
                    f32x4 sk = zero4;
                    if (st.res_mode == RES_DIRECT && st.res_tile >= 0) {   // a tensor of res_c <= Co channels (zero-padded to Co) on the same rows, in another tile
                        if (c0 < st.res_c) sk = *reinterpret_cast<const f32x4*>(lds + st.res_lds + ((1 + oy) * TWo + ox + 1) * (st.res_c + 4) + c0);
                    } else if (st.res_mode == RES_DIRECT) {
                        if (c0 < C) sk = *reinterpret_cast<const f32x4*>(tile + ((1 + oy) * TW + ox + 1) * Cs + c0);   // channels >= C: the zero pad of a widening block
                    } else if (st.res_mode == RES_MAXPOOL && c0 < (st.res_tile >= 0 ? st.res_c : C)) {
                        // 2x2 max of a tensor of twice the output's size: the stride-2 block's own input (C channels), or (res_tile) the tensor the
                        // 2x2 convolution / the stride-2 block in front of this block read (res_c channels, zero-padded to Co) — its rows 2r, 2r + 1 are still in that tile
                        const float* rt = st.res_tile >= 0 ? lds + st.res_lds : tile;
                        const int TWr = 2 * Wo + 2, Csr = st.res_tile >= 0 ? st.res_c + 4 : Cs;
                        const float* t0 = rt + ((1 + 2 * oy) * TWr + 2 * ox + 1) * Csr + c0;
                        const f32x4 s0 = *reinterpret_cast<const f32x4*>(t0), s1 = *reinterpret_cast<const f32x4*>(t0 + Csr);
                        const f32x4 s2 = *reinterpret_cast<const f32x4*>(t0 + TWr * Csr), s3 = *reinterpret_cast<const f32x4*>(t0 + TWr * Csr + Csr);
                        sk.x = fmaxf(fmaxf(s0.x, s1.x), fmaxf(s2.x, s3.x));
                        sk.y = fmaxf(fmaxf(s0.y, s1.y), fmaxf(s2.y, s3.y));
                        sk.z = fmaxf(fmaxf(s0.z, s1.z), fmaxf(s2.z, s3.z));
                        sk.w = fmaxf(fmaxf(s0.w, s1.w), fmaxf(s2.w, s3.w));
                    }
                    f32x4 v;
                    v.x = D.x + bb.x + sk.x; v.y = D.y + bb.y + sk.y; v.z = D.z + bb.z + sk.z; v.w = D.w + bb.w + sk.w;
                    v.x = fminf(fmaxf(v.x, 0.f) + sl.x * fminf(v.x, 0.f), hi);
                    v.y = fminf(fmaxf(v.y, 0.f) + sl.y * fminf(v.y, 0.f), hi);
                    v.z = fminf(fmaxf(v.z, 0.f) + sl.z * fminf(v.z, 0.f), hi);
                    v.w = fminf(fmaxf(v.w, 0.f) + sl.w * fminf(v.w, 0.f), hi);
                    if (WIDE && st.res_mode == RES_UP2X) {   // the skip joins BEHIND the activation (Epilogue::res_after); the arithmetic of block_kernels.hip's RES_UP2X
                        const int Wc = Wo >> 1;
                        const float ix = ((float)ox + 0.5f) * 0.5f - 0.5f;
                        const int x0 = max((int)floorf(ix), 0), x1 = min((int)ceilf(ix), Wc - 1);
                        const float dx = ix - (float)x0, dy = up_dy;
                        const float* c0p = dwb + c0;
                        const f32x4 p00 = *reinterpret_cast<const f32x4*>(c0p + x0 * Cso), p01 = *reinterpret_cast<const f32x4*>(c0p + x1 * Cso);
                        const f32x4 p10 = *reinterpret_cast<const f32x4*>(c0p + (Wc + x0) * Cso), p11 = *reinterpret_cast<const f32x4*>(c0p + (Wc + x1) * Cso);
                        const float w00 = (1 - dy) * (1 - dx), w10 = dy * (1 - dx), w01 = (1 - dy) * dx, w11 = dy * dx;
                        v.x += p00.x * w00 + p10.x * w10 + p01.x * w01 + p11.x * w11;
                        v.y += p00.y * w00 + p10.y * w10 + p01.y * w01 + p11.y * w11;
                        v.z += p00.z * w00 + p10.z * w10 + p01.z * w01 + p11.z * w11;
                        v.w += p00.w * w00 + p10.w * w10 + p01.w * w01 + p11.w * w11;
                    }
                    if (px < npx && c0 < Co) {
                        if (llo && (oy < st.pub_lo || oy >= nro - st.pub_hi)) {   // a row some other workgroup reads
                            const int pko = (((r0 + oy) * Wo + ox) * (Co >> 2) + (c0 >> 2)) * 32;
                            const u32x4 q0 = {__float_as_uint(v.x), tag_out, __float_as_uint(v.y), tag_out}, q1 = {__float_as_uint(v.z), tag_out, __float_as_uint(v.w), tag_out};
                            __builtin_amdgcn_raw_buffer_store_b128(q0, osrc, pko, 0, kPacketAux);
                            __builtin_amdgcn_raw_buffer_store_b128(q1, osrc, pko + 16, 0, kPacketAux);
                        }
                        if (dtile) *reinterpret_cast<f32x4*>(dtile + ((1 + oy) * TWo + ox + 1) * Cso + c0) = v;
                        if (gout) {
                            float* o = gout + (long)px * Co + c0;
                            if ((Co & 3) == 0) {
                                *reinterpret_cast<f32x4*>(o) = v;
                            } else {
                                o[0] = v.x;
                                if (c0 + 1 < Co) o[1] = v.y;
                                if (c0 + 2 < Co) o[2] = v.z;
                                if (c0 + 3 < Co) o[3] = v.w;
                            }
                        }
                    }
                }
              }
        }
        MI_BAND_STAMP(4)
        // ---- the next stage's A operands and constants: requested here, needed behind its halo rows.  (Requested in front of the depthwise phase
        // instead, into a second register set: BackCamera 178 -> 226 us, iris 182 -> 200 — the closing barriers of the phases in between wait for them.)
        const int sn = next_active(s);
        MI_BAND_STAMP(6)
        StageRegs stn{};
        if (sn < a.nstages) {
            stn = stage_regs(stage_word(lprog + sn, lane));
            MI_BAND_STAMP(7)
            fetch(stn, A, A8, A4, creg);
        }
        MI_BAND_STAMP(8)
        __syncthreads();   // the output band is complete in its tile; the input tile, the depthwise result and the constants are free
        MI_BAND_STAMP(5)
        s = sn;
        st = stn;
    }
    // ---- the last workgroup to finish moves the generation on
    if (tid == 0) {
        const unsigned total = gridDim.x * gridDim.y;
        const unsigned done = __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == total - 1) {
            __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (nobody polls any more: the next launch starts unflagged)
            __hip_atomic_store(a.sync, base / 64u + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

int bandnet_tile_floats(int R, int W, int C, int halo) { return (R + halo) * (W + 2) * (C + 4); }
int bandnet_dw_floats(const BandStage& st) { return st.kind == BAND_BLOCK ? st.R * st.Wo * (st.C + 4) : 0; }
int bandnet_const_floats(const BandStage& st) {   // what a stage stages in the constants' LDS area: bias and slopes, and the depthwise taps + biases where they fit beside them
    const int head = 32 * ((st.Co + 15) / 16), all = head + (st.kind == BAND_BLOCK ? 10 * st.C : 0);
    return all <= kConstFloats ? all : head;
}
int bandnet_lds_bytes(int tiles_floats, int dw_floats, int nstages) { return (tiles_floats + dw_floats + kConstFloats) * 4 + nstages * (int)sizeof(BandPacked) + 16; }

bool bandnet_pack(const BandStage& st, BandPacked* out) {
    auto fits = [](long v, int bits) { return v >= 0 && v < (1L << bits); };
    const long offs[8] = {st.src_off, st.dst_off, st.src_fs, st.dst_fs, st.src_ll, st.dst_ll, st.w_a, st.w_c};
    for (long v : offs)
        if (v < -1 || v > 0x7fffffffL) return false;
    if (!fits(st.kind, 1) || !fits(st.S, 2) || !fits(st.wshift, 4) || !fits(st.res_mode, 2) || !fits(st.act, 3) || !fits(st.src_tile, 2) || !fits(st.dst_tile + 1, 3) ||
        !fits(st.pub_lo, 1) || !fits(st.pub_hi, 1) || !fits(st.src_base, 3) || !fits(st.dst_base + 1, 4) || !fits(st.wpc_shift, 2) || !fits(st.res_tile + 1, 3) || !fits(st.res_c, 16) ||
        st.src_tile >= kBandTiles || st.dst_tile >= kBandTiles || st.res_tile >= kBandTiles || st.dst_base >= kBandBases || !fits(st.R, 8) || !fits(st.Rin, 8) ||
        !fits(st.dep + 1, 8) || !fits(st.nbands, 8) || !fits(st.H, 16) || !fits(st.W, 16) || !fits(st.Ho, 16) || !fits(st.Wo, 16) || !fits(st.C, 16) || !fits(st.Co, 16) ||
        !fits(st.c_floats, 12) || !fits(st.woff, 4) || st.woff >= (1 << st.wshift) || !fits(st.per_ct, 16))
        return false;
    BandPacked p{};
    if (st.pre && (st.S != 2 || st.kind != BAND_BLOCK)) return false;
    p.w[0] = (unsigned)st.kind | (unsigned)(st.pre ? 3 : st.S) << 1 | (unsigned)st.wshift << 3 | (unsigned)st.res_mode << 7 | (unsigned)st.act << 9 | (unsigned)st.src_tile << 12 |
             (unsigned)(st.dst_tile + 1) << 14 | (unsigned)st.pub_lo << 17 | (unsigned)st.pub_hi << 18 | (unsigned)st.src_base << 19 | (unsigned)(st.dst_base + 1) << 22 |
             (unsigned)st.wpc_shift << 26 | (unsigned)(st.res_tile + 1) << 28 | (unsigned)st.dst_h3 << 31;
    p.w[1] = (unsigned)st.R | (unsigned)st.Rin << 8 | (unsigned)(st.dep + 1) << 16 | (unsigned)st.nbands << 24;
    p.w[2] = (unsigned)st.H | (unsigned)st.W << 16;
    p.w[3] = (unsigned)st.Ho | (unsigned)st.Wo << 16;
    p.w[4] = (unsigned)st.C | (unsigned)st.Co << 16;
    p.w[5] = (unsigned)st.c_floats | (unsigned)st.woff << 12 | (unsigned)st.per_ct << 16;
    p.w[6] = st.mC4; p.w[7] = st.mWo; p.w[8] = st.mrowq;
    for (int k = 0; k < 8; k++) p.w[9 + k] = (unsigned)(int)offs[k];
    if (st.res_ll < -1 || st.res_ll > 0x7fffffffL || !fits(st.res_stage, 6) || !fits(st.dst_h3, 1)) return false;
    for (int off : {st.src_lds, st.dst_lds, st.res_lds})
        if (off < 0 || (off & 15) || !fits(off >> 4, 12)) return false;
    p.w[17] = (unsigned)st.res_c | (unsigned)(st.src_lds >> 4) << 16;
    p.w[18] = (unsigned)(int)st.res_ll;
    p.w[19] = (unsigned)st.res_stage | (unsigned)(st.dst_lds >> 4) << 6 | (unsigned)(st.res_lds >> 4) << 18;
    *out = p;
    return true;
}

int launch_bandnet(const BandLaunch& a, void* stream) {
    if (a.nstages < 1 || a.nstages > 63 || a.NW < 1 || a.F < 1 || a.lds_bytes > 160 * 1024 || a.tiles_floats < 0 || (a.tiles_floats & 3)) return (int)hipErrorInvalidValue;
    if (bandnet_lds_bytes(a.tiles_floats, a.dw_floats, a.nstages) > a.lds_bytes) return (int)hipErrorInvalidValue;
    if ((long)a.NW * a.F > device_cu_count()) return (int)hipErrorInvalidValue;   // every workgroup must be resident: one per CU
    auto kern = a.wide ? bandnet_kernel<false, false, true> : (a.cv2 ? bandnet_kernel<true, false, false> : (a.xb ? bandnet_kernel<false, true, false> : bandnet_kernel<false, false, false>));
    if ((a.cv2 && a.xb) || (a.wide && (a.cv2 || a.xb))) return (int)hipErrorInvalidValue;   // (no such program: the planner does not build one)
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
#ifdef MI_BAND_STAMPS
    BandLaunch b = a;
    b.stamps = g_band_stamps;
    return (int)launch_kernel(kern, dim3((unsigned)a.NW, (unsigned)a.F), dim3(kThreads), (size_t)a.lds_bytes, (hipStream_t)stream, b);
#else
    return (int)launch_kernel(kern, dim3((unsigned)a.NW, (unsigned)a.F), dim3(kThreads), (size_t)a.lds_bytes, (hipStream_t)stream, a);
#endif
}

}  // namespace mi

#ifdef MI_BAND_STAMPS
// stamps build only (tools/band_stamps.py)
extern "C" void mi_debug_set_band_stamps(unsigned long long* p) { mi::g_band_stamps = p; }
#endif
