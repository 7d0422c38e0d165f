// chain_kernels.hip — frame-resident chains of BlazeBlocks for the small-spatial, channel-heavy stages.
//
// At 16x16x96 (BackCamera, 7 consecutive blocks), 12x12x128 / 6x6x128 (face mesh) … a whole frame fits in the 160 KB LDS
// of a CU, while per-block launches are latency-bound (few pixels, 144 dependent MFMAs per 32-pixel group).  Here ONE
// 512-thread workgroup loads a frame into LDS once, runs every block of the chain on it
//     x <- act( PW1x1( DW3x3(x) + b_dw ) + b_pw + x )
// and writes the frame back once: the chain's intermediate activations never touch HBM (6 of 7 round trips removed for
// the 16x16 stage) and 7 launches become 1.  (Replaces the same TFLite op chains as block_kernels.hip; reference call
// site /root/reference/src/face_detection_lite/face_detection.rs:235.)
//
//   * each of the 8 waves owns one 32-pixel group of the frame for the whole chain (H*W <= 256);
//   * per block: depthwise 3x3 on the VALU in the MFMA B-operand layout (lane = pixel x k-half, see block_kernels.hip),
//     v_mfma_f32_32x32x2_f32 over all output-channel tiles, the MFMAs of channel chunk j interleaved with the depthwise
//     math of chunk j+1; pointwise weights stream from L2 in A-fragment order, one chunk ahead;
//   * epilogue values stay in registers across a workgroup barrier (all reads of x done), then overwrite x in place.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "kernels.hpp"
#include "launch.hpp"

namespace mi {

typedef float f32x16c __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ float4 cld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct ChainGeom {
    int Cp, Ch, C4, PS, RS, MT;
    int off_wdw, off_bdw, off_bias, off_alpha;  // LDS offsets (floats)
    int split;     // stride-1 blocks run as (32-pixel group, output tile) units, one per wave
    int pre_P, pre_RP, pre_GP, pre_PS, pre_RS;  // `pre`: passes, output rows per pass, 32-pixel groups per pass, staged pixel / row stride
    int off_t8;    // where `post`'s output is kept for the heads that read it ([pixels][post.Co + 4]); -1: not kept
    int lds_bytes;
    unsigned long long* stamps;  // diagnostic builds only (MI_CHAIN_STAMPS): 24 s_memtime stamps per workgroup
};
#ifdef MI_CHAIN_STAMPS
unsigned long long* g_chain_stamps = nullptr;
#define MI_CHAIN_STAMP(k) if (g.stamps && threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); g.stamps[(long)blockIdx.x * 24 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define MI_CHAIN_STAMP(k)
#endif

// SPLIT: frames of so few 32-pixel groups that every stage runs as (group, output tile) units, one per wave (g.split)
template <int MT, bool SPLIT>
__global__ __launch_bounds__(512, 2) void chain_kernel(ChainArgs a, ChainGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tile = lds;  // [(H+2)][(W+2)][PS], zero border
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const float* in = a.in + (long)b * a.in_fs;
    const int rowf4 = a.W * g.C4;
    const int nch = g.Ch >> 2;

    // small per-block constants -> LDS (depthwise taps [9][Cs] zero-padded to Cp, depthwise bias, pointwise bias, negative slopes).
    // fetch_consts issues the global loads of a stage's constants into registers BEFORE the previous stage computes; commit_consts
    // writes them to LDS after the barrier that ends that stage: the L2 round trip hides under the stage's MFMAs.
    struct ConstRegs { float w[3], bdw, bias, alpha; };
    auto fetch_consts = [&](const ChainBlock& cb, int Cs, int Cos, ConstRegs& r) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = tid + 512 * k, c = i % g.Cp;
            r.w[k] = (i < 9 * g.Cp && c < Cs) ? cb.w_dw[(i / g.Cp) * Cs + c] : 0.f;
        }
        r.bdw = (tid < Cs && cb.b_dw) ? cb.b_dw[tid] : 0.f;
        r.bias = (tid < Cos && cb.bias) ? cb.bias[tid] : 0.f;
        r.alpha = (tid < Cos && cb.act == ACT_PRELU) ? cb.alpha[tid] : (cb.act == ACT_NONE ? 1.f : 0.f);
    };
    auto commit_consts = [&](const ConstRegs& r) {
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (tid + 512 * k < 9 * g.Cp) lds[g.off_wdw + tid + 512 * k] = r.w[k];
        if (tid < g.Cp) lds[g.off_bdw + tid] = r.bdw;
        if (tid < MT * 32) { lds[g.off_bias + tid] = r.bias; lds[g.off_alpha + tid] = r.alpha; }
    };
    // pointwise weights, packed [tile][chunk][lane][4] in global memory (L2)
    auto a_frag = [&](const ChainBlock& cb, int nchk, int j, float4 (&av)[MT]) {
#ifdef MI_ABL_CHAIN_NOA  // timing ablation (development only): no weight loads inside the contraction
#pragma unroll
        for (int m = 0; m < MT; m++) { av[m] = make_float4(1.f, 2.f, 3.f, 4.f); asm volatile("" : "+v"(av[m].x), "+v"(av[m].y), "+v"(av[m].z), "+v"(av[m].w)); }
#else
#pragma unroll
        for (int m = 0; m < MT; m++) av[m] = cld4(cb.w_pw + (((long)m * nchk + j) * 64 + lane) * 4);
#endif
    };
    auto mfma_chunk = [&](const float4 (&av)[MT], const float4& bf, auto& D) {
#ifdef MI_ABL_CHAIN_NOMFMA  // timing ablation: one VALU op per chunk and tile instead of the four MFMAs
#pragma unroll
        for (int m = 0; m < MT; m++) D[m][0] += av[m].x * bf.x + av[m].y * bf.y + av[m].z * bf.z + av[m].w * bf.w;
        return;
#endif
#pragma unroll
        for (int m = 0; m < MT; m++) {
            D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].x, bf.x, D[m], 0, 0, 0);
            D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].y, bf.y, D[m], 0, 0, 0);
            D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].z, bf.z, D[m], 0, 0, 0);
            D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].w, bf.w, D[m], 0, 0, 0);
        }
    };
    // One 32-pixel group on this wave: depthwise chunk j (4 channels of this lane's k-half) by `dw`, the MFMAs of chunk j
    // interleaved with the depthwise math of chunk j + 1, pointwise weights streamed from L2 one chunk ahead.
    auto contract = [&](const ChainBlock& cb, int nchk, auto&& dw, auto& D) {
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) D[m][e] = 0.f;
        float4 bf, av[MT];
        dw(0, bf);
        a_frag(cb, nchk, 0, av);
        for (int j = 0; j + 1 < nchk; j++) {
            float4 bn, an[MT];
            mfma_chunk(av, bf, D);
            a_frag(cb, nchk, j + 1, an);
            dw(j + 1, bn);
            bf = bn;
#pragma unroll
            for (int m = 0; m < MT; m++) av[m] = an[m];
            constexpr int NM = 4 * MT;
#pragma unroll
            for (int k = 0; k < NM; k++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, (19 + NM - 1) / NM, 0);  // DS read
                __builtin_amdgcn_sched_group_barrier(0x002, (20 + NM - 1) / NM, 0);  // VALU
            }
        }
        mfma_chunk(av, bf, D);
    };
    // ONE output tile (mt) of a 32-pixel group: the unit of work when a stage has so few pixels that whole groups would leave
    // waves idle (the 8x8 / 6x6 frames, the stride-2 block behind the chain, the output heads).  A unit is only 4 nchk dependent
    // MFMAs, too short to hide an L2 round trip per chunk behind the previous chunk: every weight chunk is in flight up front.
    constexpr int NJ = MT * 4;  // K <= 32 MT  ->  nchk <= 4 MT
    auto contract1 = [&](const float* w_pw, int nchk, int mt, auto&& dw, f32x16c& D1) {
        const float* wa = w_pw + ((long)mt * nchk * 64 + lane) * 4;
        float4 aw[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++)
            if (j < nchk) aw[j] = cld4(wa + 256 * j);
#pragma unroll
        for (int e = 0; e < 16; e++) D1[e] = 0.f;
        float4 bf;
        dw(0, bf);
#pragma unroll
        for (int j = 0; j < NJ; j++)
            if (j < nchk) {
                float4 bn = bf;
                // the chunk number reaches `dw` through an opaque scalar: with a literal, every tap address of every chunk is
                // loop-invariant and gets hoisted out of the unit loop into a VGPR of its own (spills)
                int jn = j + 1;
                asm volatile("" : "+s"(jn));
                if (j + 1 < nchk) dw(jn, bn);
                D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j].x, bf.x, D1, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j].y, bf.y, D1, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j].z, bf.z, D1, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j].w, bf.w, D1, 0, 0, 0);
                bf = bn;
                __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from hoisting every chunk's LDS reads to the top (spills)
            }
    };
    // bias (+ skip) + activation of the 4 channels ch .. ch + 3 (register quad gq) of this lane, in place
    auto finish1 = [&](f32x16c& D1, int ch, int gq, const float4& skip, float hi) {
        const float4 bb = cld4(lds + g.off_bias + ch), al = cld4(lds + g.off_alpha + ch);
        const float4 v = make_float4(D1[4 * gq] + bb.x + skip.x, D1[4 * gq + 1] + bb.y + skip.y, D1[4 * gq + 2] + bb.z + skip.z, D1[4 * gq + 3] + bb.w + skip.w);
        D1[4 * gq] = fminf(fmaxf(v.x, 0.f) + al.x * fminf(v.x, 0.f), hi);
        D1[4 * gq + 1] = fminf(fmaxf(v.y, 0.f) + al.y * fminf(v.y, 0.f), hi);
        D1[4 * gq + 2] = fminf(fmaxf(v.z, 0.f) + al.z * fminf(v.z, 0.f), hi);
        D1[4 * gq + 3] = fminf(fmaxf(v.w, 0.f) + al.w * fminf(v.w, 0.f), hi);
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // this wave's pixel group in the stride-1 blocks; g.split: frames of so few groups that (group, output tile) units fit the 8 waves
    const int ngrp = (a.H * a.W + 31) >> 5;
    const int grp_w = SPLIT ? wave % ngrp : wave, mt_w = SPLIT ? wave / ngrp : 0;
    const int q = grp_w * 32 + pl;
    const bool valid = q < a.H * a.W;
    const int oy = valid ? q / a.W : 0, ox = valid ? q - (q / a.W) * a.W : 0;
    const bool wave_active = SPLIT ? wave < ngrp * MT : wave * 32 < a.H * a.W;
    const int base0 = oy * g.RS + ox * g.PS + h * g.Ch;  // tap (ky, kx) = base0 + ky*RS + kx*PS (input row oy-1+ky at slot oy+ky)
    float* centre = tile + (oy + 1) * g.RS + (ox + 1) * g.PS;

    MI_CHAIN_STAMP(0)
    ConstRegs cr;
    if (a.pre.on) {
        // ---- the stride-2 block in front of the chain: its (2H x 2W x Cin) input is staged through the (still empty) tile region in
        // g.pre_P passes of g.pre_RP output rows — coalesced loads, every byte read once — and its 3x3 stride-2 taps and 2x2 max-pool
        // skip are gathered from LDS.  (Gathering them from global memory, 16 bytes per lane and tap, cost one cache-line lookup per
        // lane: 60 us of a 110 us launch.)  Pass p runs on waves [p GP, (p + 1) GP): a wave computes in at most one pass and keeps its
        // result in registers until the staged input is dead; then the tile is cleared and the results become the resident frame.
        const ChainBlock& cb = a.pre.blk;
        const int Cin = a.pre.Cin, Chp = Cin >> 1, Hi = 2 * a.H, Wi = 2 * a.W, C4i = Cin >> 2;
        const int PSi = g.pre_PS, RSi = g.pre_RS, rowf4i = Wi * C4i;
        const float* src = a.pre.in + (long)b * a.pre.in_fs;
        fetch_consts(cb, Cin, a.C, cr);
        commit_consts(cr);
        fetch_consts(a.blocks[0], a.C, a.C, cr);  // block 0's constants travel while this stage runs
        const float* wdw = lds + g.off_wdw;
        const float* bdw = lds + g.off_bdw;
        f32x16c D[SPLIT ? 1 : MT];
        bool mine = false;
        float* my_centre = tile;
        int my_mt = 0;
        const int UP = SPLIT ? g.pre_GP * MT : g.pre_GP;  // units (SPLIT) or whole groups per pass
        for (int p = 0; p < g.pre_P; p++) {
            const int r0 = p * g.pre_RP, r1 = min(r0 + g.pre_RP, a.H);
            const int nrows = 2 * (r1 - r0) + 1;  // input rows 2 r0 .. 2 r1 (row Hi and column Wi are the SAME padding: zero)
            if (p) __syncthreads();               // the previous pass has read its rows
            for (int i = tid; i < nrows * rowf4i; i += 512) {
                const int rr = i / rowf4i, e = i - rr * rowf4i, px = e / C4i, c4 = e - px * C4i, iy = 2 * r0 + rr;
                *reinterpret_cast<float4*>(lds + rr * RSi + px * PSi + 4 * c4) = iy < Hi ? cld4(src + ((long)iy * rowf4i + e) * 4) : zero4;
            }
            for (int i = tid; i < nrows * C4i; i += 512) {
                const int rr = i / C4i, c4 = i - rr * C4i;
                *reinterpret_cast<float4*>(lds + rr * RSi + Wi * PSi + 4 * c4) = zero4;
            }
            __syncthreads();
            MI_CHAIN_STAMP(16 + p)
            const int u = wave - p * UP;
            if (u >= 0 && u < UP) {  // wave-uniform
                const int ugrp = SPLIT ? u % g.pre_GP : u, umt = SPLIT ? u / g.pre_GP : 0;
                const int qp = ugrp * 32 + pl;
                const bool vp = qp < (r1 - r0) * a.W;
                const int oyl = vp ? qp / a.W : 0, oxp = vp ? qp - (qp / a.W) * a.W : 0;
                const float* t0 = lds + (2 * oyl) * RSi + (2 * oxp) * PSi;  // tap (ky, kx) of this lane's output pixel
                auto dw = [&](int j, float4& bf) {
                    const int c0 = h * Chp + 4 * j;
                    bf = cld4(bdw + c0);
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            const float4 w = cld4(wdw + (ky * 3 + kx) * g.Cp + c0);
                            const float4 d = cld4(t0 + ky * RSi + kx * PSi + c0);
                            bf.x = fmaf(d.x, w.x, bf.x);
                            bf.y = fmaf(d.y, w.y, bf.y);
                            bf.z = fmaf(d.z, w.z, bf.z);
                            bf.w = fmaf(d.w, w.w, bf.w);
                        }
                };
                if constexpr (SPLIT) contract1(cb.w_pw, Chp >> 2, umt, dw, D[0]);
                else contract(cb, Chp >> 2, dw, D);
                const float hi = cb.act == ACT_RELU6 ? 6.f : INFINITY;
#pragma unroll
                for (int m = 0; m < (SPLIT ? 1 : MT); m++)
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const int ch = (SPLIT ? umt : m) * 32 + 8 * gq + 4 * h;
                        if (ch >= a.C) continue;
                        float4 sk = zero4;
                        if (cb.has_res && ch < Cin) {  // 2x2 max-pool of the input, channels above Cin are the zero pad
                            const float4 s0 = cld4(t0 + ch), s1 = cld4(t0 + PSi + ch), s2 = cld4(t0 + RSi + ch), s3 = cld4(t0 + RSi + PSi + ch);
                            sk = make_float4(fmaxf(fmaxf(s0.x, s1.x), fmaxf(s2.x, s3.x)), fmaxf(fmaxf(s0.y, s1.y), fmaxf(s2.y, s3.y)),
                                             fmaxf(fmaxf(s0.z, s1.z), fmaxf(s2.z, s3.z)), fmaxf(fmaxf(s0.w, s1.w), fmaxf(s2.w, s3.w)));
                        }
                        finish1(D[m], ch, gq, sk, hi);
                    }
                mine = vp;
                my_mt = umt;
                my_centre = tile + (r0 + oyl + 1) * g.RS + (oxp + 1) * g.PS;
            }
        }
        __syncthreads();  // the staged input is dead
        for (int i = tid; i < ((a.H + 2) * g.RS) >> 2; i += 512) reinterpret_cast<float4*>(tile)[i] = zero4;
        commit_consts(cr);
        __syncthreads();
        if (mine) {
#pragma unroll
            for (int m = 0; m < (SPLIT ? 1 : MT); m++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int ch = (SPLIT ? my_mt : m) * 32 + 8 * gq + 4 * h;
                    if (ch < a.C) *reinterpret_cast<float4*>(my_centre + ch) = make_float4(D[m][4 * gq], D[m][4 * gq + 1], D[m][4 * gq + 2], D[m][4 * gq + 3]);
                }
        }
    } else {
        // ---- load the frame (coalesced)
        fetch_consts(a.blocks[0], a.C, a.C, cr);
        for (int i = tid; i < ((a.H + 2) * g.RS) >> 2; i += 512) reinterpret_cast<float4*>(tile)[i] = zero4;
        __syncthreads();
        for (int i = tid; i < a.H * rowf4; i += 512) {
            int r = i / rowf4, e = i - r * rowf4;
            int px = e / g.C4, c4 = e - px * g.C4;
            *reinterpret_cast<float4*>(tile + (r + 1) * g.RS + (px + 1) * g.PS + 4 * c4) = cld4(in + 4 * (long)i);
        }
        commit_consts(cr);
    }
    __syncthreads();

    MI_CHAIN_STAMP(1)
    for (int blk = 0; blk < a.nblocks; blk++) {
        const ChainBlock& cb = a.blocks[blk];
        // ---- the next stage's small constants start their trip now (committed to LDS behind this block's barrier)
        const bool more = blk + 1 < a.nblocks;
        if (more) fetch_consts(a.blocks[blk + 1], a.C, a.C, cr);
        else if (a.post.on) fetch_consts(a.post.blk, a.C, a.post.Co, cr);
        const float* wdw = lds + g.off_wdw;
        const float* bdw = lds + g.off_bdw;

        f32x16c D[SPLIT ? 1 : MT];
        if (wave_active) {
            auto dw = [&](int j, float4& bf) {
                const float* wj = wdw + h * g.Ch + 4 * j;
                bf = zero4;
#ifdef MI_ABL_CHAIN_NODW  // timing ablation: centre tap only
                bf = cld4(tile + base0 + g.RS + g.PS + 4 * j);
                return;
#endif
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        const float4 w = cld4(wj + (ky * 3 + kx) * g.Cp);
                        const float4 d = cld4(tile + base0 + ky * g.RS + kx * g.PS + 4 * j);
                        bf.x = fmaf(d.x, w.x, bf.x);
                        bf.y = fmaf(d.y, w.y, bf.y);
                        bf.z = fmaf(d.z, w.z, bf.z);
                        bf.w = fmaf(d.w, w.w, bf.w);
                    }
                const float4 bb = cld4(bdw + h * g.Ch + 4 * j);
                bf.x += bb.x; bf.y += bb.y; bf.z += bb.z; bf.w += bb.w;
            };
            // ---- contraction, then the epilogue into registers (reads x at the centre pixel), written back after the barrier
            const float hi = cb.act == ACT_RELU6 ? 6.f : INFINITY;
            if constexpr (SPLIT) contract1(cb.w_pw, nch, mt_w, dw, D[0]);
            else contract(cb, nch, dw, D);
#pragma unroll
            for (int m = 0; m < (SPLIT ? 1 : MT); m++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int ch = (SPLIT ? mt_w : m) * 32 + 8 * gq + 4 * h;
                    if (ch < a.C) finish1(D[m], ch, gq, cb.has_res ? cld4(centre + ch) : zero4, hi);
                }
        }
        __syncthreads();  // every wave has read x (and the constants) for this block
        if (more || a.post.on) commit_consts(cr);
        if (wave_active && valid) {
#pragma unroll
            for (int m = 0; m < (SPLIT ? 1 : MT); m++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int ch = (SPLIT ? mt_w : m) * 32 + 8 * gq + 4 * h;
                    if (ch < a.C) *reinterpret_cast<float4*>(centre + ch) = make_float4(D[m][4 * gq], D[m][4 * gq + 1], D[m][4 * gq + 2], D[m][4 * gq + 3]);
                }
        }
        __syncthreads();
        MI_CHAIN_STAMP(2 + blk)
    }
    // ---- write the frame back (coalesced 16 B per lane, consecutive addresses)
    if (a.write_out) {
        float* out = a.out + (long)b * a.out_fs;
        for (int i = tid; i < a.H * rowf4; i += 512) {
            int r = i / rowf4, e = i - r * rowf4;
            int px = e / g.C4, c4 = e - px * g.C4;
            *reinterpret_cast<float4*>(out + 4 * (long)i) = cld4(tile + (r + 1) * g.RS + (px + 1) * g.PS + 4 * c4);
        }
    }
    MI_CHAIN_STAMP(10)
    int post_units = 0;
    if (a.post.on) {
        // ---- the stride-2 block behind the chain: taps from the resident frame, output straight to global memory
        const ChainBlock& cb = a.post.blk;
        const int Ho = a.H >> 1, Wo = a.W >> 1, Co = a.post.Co, npo = Ho * Wo;  // constants: committed behind the last block
        const float* wdw = lds + g.off_wdw;
        const float* bdw = lds + g.off_bdw;
        const int ngo = (npo + 31) >> 5, MTo = (Co + 31) >> 5;
        post_units = ngo * MTo;
        for (int u = wave; u < post_units; u += 8) {  // wave-uniform; unit = (32 output pixels, 32 output channels)
            const int grp = u % ngo, mt = u / ngo;
            const int qo = grp * 32 + pl;
            const bool vo = qo < npo;
            const int py = vo ? qo / Wo : 0, px = vo ? qo - (qo / Wo) * Wo : 0;
            // SAME on an even size: taps at image rows 2py .. 2py+2 = tile slots 2py+1 .. 2py+3 (slot H+1 is the zero border)
            const float* t0 = tile + (2 * py + 1) * g.RS + (2 * px + 1) * g.PS;
            f32x16c D1;
            auto dw = [&](int j, float4& bf) {
                const int c0 = h * g.Ch + 4 * j;
                bf = cld4(bdw + c0);
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        const float4 w = cld4(wdw + (ky * 3 + kx) * g.Cp + c0);
                        const float4 d = cld4(t0 + ky * g.RS + kx * g.PS + c0);
                        bf.x = fmaf(d.x, w.x, bf.x);
                        bf.y = fmaf(d.y, w.y, bf.y);
                        bf.z = fmaf(d.z, w.z, bf.z);
                        bf.w = fmaf(d.w, w.w, bf.w);
                    }
            };
            contract1(cb.w_pw, nch, mt, dw, D1);
            const float hi = cb.act == ACT_RELU6 ? 6.f : INFINITY;
            float* dst = a.post.out + (long)b * a.post.out_fs + (long)qo * Co;
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const int ch = mt * 32 + 8 * gq + 4 * h;
                if (ch >= Co) continue;
                float4 sk = zero4;
                if (cb.has_res && ch < a.C) {  // 2x2 max-pool of the resident frame, zero channel pad above C
                    const float4 s0 = cld4(t0 + ch), s1 = cld4(t0 + g.PS + ch), s2 = cld4(t0 + g.RS + ch), s3 = cld4(t0 + g.RS + g.PS + ch);
                    sk = make_float4(fmaxf(fmaxf(s0.x, s1.x), fmaxf(s2.x, s3.x)), fmaxf(fmaxf(s0.y, s1.y), fmaxf(s2.y, s3.y)),
                                     fmaxf(fmaxf(s0.z, s1.z), fmaxf(s2.z, s3.z)), fmaxf(fmaxf(s0.w, s1.w), fmaxf(s2.w, s3.w)));
                }
                finish1(D1, ch, gq, sk, hi);
                if (vo) {
                    const float4 v = make_float4(D1[4 * gq], D1[4 * gq + 1], D1[4 * gq + 2], D1[4 * gq + 3]);
                    *reinterpret_cast<float4*>(dst + ch) = v;
                    if (g.off_t8 >= 0) *reinterpret_cast<float4*>(lds + g.off_t8 + qo * (Co + 4) + ch) = v;
                }
            }
        }
    }
    MI_CHAIN_STAMP(11)
    // ---- output heads: stacked 1x1 convolutions on the resident frame / on `post`'s output; unit = (32 pixels, 32 stacked rows)
    for (int hd = 0; hd < kChainHeads; hd++) {
        MI_CHAIN_STAMP(12 + hd)
        const ChainHead& H = a.heads[hd];
        if (!H.on) continue;
        if (H.src == 1) { __syncthreads(); post_units = 0; }  // `post`'s LDS copy is complete; heads on the frame itself need no barrier
        const bool from_post = H.src == 1;
        const int Wh = from_post ? a.W >> 1 : a.W, np = from_post ? (a.H >> 1) * (a.W >> 1) : a.H * a.W;
        const int Cs = from_post ? a.post.Co : a.C, Chs = Cs >> 1, nchs = Chs >> 2;
        const int Cot = H.Co_a + H.Co_b, MTh = (Cot + 31) >> 5, ng = (np + 31) >> 5;
        // heads on the frame start on the waves that `post` left without a unit
        for (int u = (wave - post_units) & 7; u < ng * MTh; u += 8) {  // wave-uniform
            const int grp = u % ng, mt = u / ng;
            const int qh = grp * 32 + pl;
            const bool vh = qh < np;
            const int qc = vh ? qh : 0;
            const float* px = from_post ? lds + g.off_t8 + qc * (Cs + 4) : tile + (qc / Wh + 1) * g.RS + (qc % Wh + 1) * g.PS;
            float4 bq[4];
#pragma unroll
            for (int gq = 0; gq < 4; gq++) bq[gq] = H.bias ? cld4(H.bias + mt * 32 + 8 * gq + 4 * h) : zero4;  // stacked, zero padded to the tile
            f32x16c D1;
            contract1(H.w_pw, nchs, mt, [&](int j, float4& bf) { bf = cld4(px + h * Chs + 4 * j); }, D1);
            if (!vh) continue;
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const int ch = mt * 32 + 8 * gq + 4 * h;   // stacked row
                if (ch >= Cot) continue;
                const float v[4] = {D1[4 * gq] + bq[gq].x, D1[4 * gq + 1] + bq[gq].y, D1[4 * gq + 2] + bq[gq].z, D1[4 * gq + 3] + bq[gq].w};
                if (ch < H.Co_a) {  // Co_a % 4 == 0: the quad lies in one head
                    *reinterpret_cast<float4*>(H.out_a + (long)b * H.out_a_fs + (long)qh * H.Co_a + ch) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    float* ob = H.out_b + (long)b * H.out_b_fs + (long)qh * H.Co_b + (ch - H.Co_a);
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (ch + e < Cot) ob[e] = v[e];
                }
            }
        }
    }
    MI_CHAIN_STAMP(14)
}

bool make_chain_geom(const ChainArgs& a, ChainGeom* out) {
    ChainGeom g{};
    if (a.C % 8 || a.H * a.W > 256 || a.nblocks < 1 || a.nblocks > kMaxChain) return false;
    g.Cp = a.C; g.Ch = a.C / 2; g.C4 = a.C / 4; g.PS = a.C + 4; g.RS = (a.W + 2) * g.PS;
    g.MT = (a.C + 31) / 32;
    if (g.MT > 4) return false;
    g.split = g.MT > 1 && ((a.H * a.W + 31) / 32) * g.MT <= 8;
    int off = (a.H + 2) * g.RS;
    if (a.pre.on) {
        // staging area of `pre`'s input (shares the tile region, grows it when larger): fewest passes whose rows fit and whose
        // 32-pixel groups all find a wave of their own
        if (a.pre.Cin % 8 || a.pre.Cin < 8 || a.pre.Cin > a.C) return false;
        g.pre_PS = a.pre.Cin + 4;
        g.pre_RS = (2 * a.W + 1) * g.pre_PS;
        g.pre_P = 0;
        for (int P = 1; P <= 4 && !g.pre_P; P++) {
            const int RP = (a.H + P - 1) / P, GP = (RP * a.W + 31) / 32;
            const long stage = (long)(2 * RP + 1) * g.pre_RS;
            const long rest = 9 * g.Cp + g.Cp + 8 * 32 + 64 + (a.post.on ? (long)(a.H >> 1) * (a.W >> 1) * (a.post.Co + 4) : 0);
            if ((P - 1) * RP < a.H && P * GP * (g.split ? g.MT : 1) <= 8 && (std::max<long>(stage, off) + rest) * 4 <= 160 * 1024 - 256) { g.pre_P = P; g.pre_RP = RP; g.pre_GP = GP; off = (int)std::max<long>(stage, off); }
        }
        if (!g.pre_P) return false;
    }
    g.off_wdw = off; off += 9 * g.Cp;
    g.off_bdw = off; off += g.Cp;
    off = (off + 3) & ~3;
    g.off_bias = off; off += g.MT * 32;
    g.off_alpha = off; off += g.MT * 32;
    g.off_t8 = -1;
    for (int hd = 0; hd < kChainHeads; hd++) {
        const ChainHead& H = a.heads[hd];
        if (!H.on) continue;
        if (H.Co_a < 4 || H.Co_a % 4 || H.Co_b < 0 || H.Co_a + H.Co_b > 128 || !H.out_a || (H.Co_b && !H.out_b) || !H.w_pw) return false;
        if ((reinterpret_cast<uintptr_t>(H.out_a) & 15) || (H.out_a_fs & 3) || (reinterpret_cast<uintptr_t>(H.w_pw) & 15)) return false;
        if (H.src == 1) {
            if (!a.post.on) return false;
            if (g.off_t8 < 0) { g.off_t8 = off; off += (a.H >> 1) * (a.W >> 1) * (a.post.Co + 4); }
        }
    }
    g.lds_bytes = off * 4;
    if (g.lds_bytes > 160 * 1024 - 256) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!aligned16(a.in) || !aligned16(a.out) || (a.in_fs & 3) || (a.out_fs & 3)) return false;
    if (a.pre.on && (a.pre.Cin % 8 || a.pre.Cin < 8 || a.pre.Cin > a.C || !aligned16(a.pre.in) || (a.pre.in_fs & 3))) return false;
    if (a.post.on && ((a.H & 1) || (a.W & 1) || a.post.Co % 4 || a.post.Co < a.C || a.post.Co > g.MT * 32 || !aligned16(a.post.out) || (a.post.out_fs & 3))) return false;
    *out = g;
    return true;
}

template <int MT, bool SPLIT>
int launch_chain_inst(const ChainArgs& a, const ChainGeom& g, hipStream_t s) {
    auto kern = chain_kernel<MT, SPLIT>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    return (int)launch_kernel(kern, dim3((unsigned)a.B), dim3(512), (size_t)g.lds_bytes, s, a, g);
}

}  // namespace

bool chain_kernel_supports(const ChainArgs& a) {
    ChainGeom g;
    return make_chain_geom(a, &g);
}

int launch_chain(const ChainArgs& a, void* stream) {
    ChainGeom g;
    if (!make_chain_geom(a, &g)) return (int)hipErrorInvalidValue;
#ifdef MI_CHAIN_STAMPS
    g.stamps = g_chain_stamps;
#endif
    hipStream_t s = (hipStream_t)stream;
    switch (g.MT) {
        case 1: return launch_chain_inst<1, false>(a, g, s);
        case 2: return g.split ? launch_chain_inst<2, true>(a, g, s) : launch_chain_inst<2, false>(a, g, s);
        case 3: return g.split ? launch_chain_inst<3, true>(a, g, s) : launch_chain_inst<3, false>(a, g, s);
        case 4: return g.split ? launch_chain_inst<4, true>(a, g, s) : launch_chain_inst<4, false>(a, g, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace mi

#ifdef MI_CHAIN_STAMPS
// stamps build only (tools/chain_stamps.py)
extern "C" void mi_debug_set_chain_stamps(unsigned long long* p) { mi::g_chain_stamps = p; }
#endif
