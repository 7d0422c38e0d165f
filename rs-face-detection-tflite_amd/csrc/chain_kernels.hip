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
    int off_t8;    // where `post`'s output is kept for the heads that read it ([pixels][post.Co + 4]); -1: not kept
    int lds_bytes;
};

template <int MT>
__global__ __launch_bounds__(512, 2) void chain_kernel(ChainArgs a, ChainGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tile = lds;  // [(H+2)][(W+2)][PS], zero border
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const float* in = a.in + (long)b * a.in_fs;
    const int rowf4 = a.W * g.C4;
    const int nch = g.Ch >> 2;

    // small per-block constants -> LDS (depthwise taps [9][Cs] zero-padded to Cp, depthwise bias, pointwise bias, negative slopes)
    auto stage_consts = [&](const ChainBlock& cb, int Cs, int Cos) {
        for (int i = tid; i < 9 * g.Cp; i += 512) {
            int c = i % g.Cp;
            lds[g.off_wdw + i] = c < Cs ? cb.w_dw[(i / g.Cp) * Cs + c] : 0.f;
        }
        for (int i = tid; i < g.Cp; i += 512) lds[g.off_bdw + i] = (i < Cs && cb.b_dw) ? cb.b_dw[i] : 0.f;
        for (int i = tid; i < MT * 32; i += 512) {
            lds[g.off_bias + i] = (i < Cos && cb.bias) ? cb.bias[i] : 0.f;
            lds[g.off_alpha + i] = (i < Cos && cb.act == ACT_PRELU) ? cb.alpha[i] : (cb.act == ACT_NONE ? 1.f : 0.f);
        }
    };
    // One 32-pixel group on this wave: depthwise chunk j (4 channels of this lane's k-half) by `dw`, the MFMAs of chunk j
    // interleaved with the depthwise math of chunk j + 1, pointwise weights streamed from L2 one chunk ahead.
    auto contract = [&](const ChainBlock& cb, int nchk, auto&& dw, f32x16c (&D)[MT]) {
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) D[m][e] = 0.f;
        auto a_frag = [&](int j, float4 (&av)[MT]) {  // packed [tile][chunk][lane][4] in global/L2
#pragma unroll
            for (int m = 0; m < MT; m++) av[m] = cld4(cb.w_pw + (((long)m * nchk + j) * 64 + lane) * 4);
        };
        auto mfma_chunk = [&](const float4 (&av)[MT], const float4& bf) {
#pragma unroll
            for (int m = 0; m < MT; m++) {
                D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].x, bf.x, D[m], 0, 0, 0);
                D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].y, bf.y, D[m], 0, 0, 0);
                D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].z, bf.z, D[m], 0, 0, 0);
                D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].w, bf.w, D[m], 0, 0, 0);
            }
        };
        float4 bf, av[MT];
        dw(0, bf);
        a_frag(0, av);
        for (int j = 0; j + 1 < nchk; j++) {
            float4 bn, an[MT];
            mfma_chunk(av, bf);
            a_frag(j + 1, an);
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
        mfma_chunk(av, bf);
    };
    // bias (+ skip) + activation of the 4 channels (m, gq) of this lane, in place in D
    auto finish = [&](f32x16c (&D)[MT], int m, int gq, const float4& skip, float hi) {
        const int ch = m * 32 + 8 * gq + 4 * h;
        const float4 bb = cld4(lds + g.off_bias + ch), al = cld4(lds + g.off_alpha + ch);
        const float4 v = make_float4(D[m][4 * gq] + bb.x + skip.x, D[m][4 * gq + 1] + bb.y + skip.y, D[m][4 * gq + 2] + bb.z + skip.z, D[m][4 * gq + 3] + bb.w + skip.w);
        D[m][4 * gq] = fminf(fmaxf(v.x, 0.f) + al.x * fminf(v.x, 0.f), hi);
        D[m][4 * gq + 1] = fminf(fmaxf(v.y, 0.f) + al.y * fminf(v.y, 0.f), hi);
        D[m][4 * gq + 2] = fminf(fmaxf(v.z, 0.f) + al.z * fminf(v.z, 0.f), hi);
        D[m][4 * gq + 3] = fminf(fmaxf(v.w, 0.f) + al.w * fminf(v.w, 0.f), hi);
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // this wave's pixel group
    const int q = wave * 32 + pl;
    const bool valid = q < a.H * a.W;
    const int oy = valid ? q / a.W : 0, ox = valid ? q - (q / a.W) * a.W : 0;
    const bool wave_active = wave * 32 < a.H * a.W;
    const int base0 = oy * g.RS + ox * g.PS + h * g.Ch;  // tap (ky, kx) = base0 + ky*RS + kx*PS (input row oy-1+ky at slot oy+ky)
    float* centre = tile + (oy + 1) * g.RS + (ox + 1) * g.PS;

    for (int i = tid; i < ((a.H + 2) * g.RS) >> 2; i += 512) reinterpret_cast<float4*>(tile)[i] = zero4;
    if (a.pre.on) {
        // ---- the stride-2 block in front of the chain: its (2H x 2W x Cin) input comes straight from global memory (L2 / MALL),
        // its output becomes the resident frame
        const ChainBlock& cb = a.pre.blk;
        const int Cin = a.pre.Cin, Chp = Cin >> 1, Hi = 2 * a.H, Wi = 2 * a.W;
        stage_consts(cb, Cin, a.C);
        __syncthreads();
        if (wave_active) {
            const float* src = a.pre.in + (long)b * a.pre.in_fs;
            const float* wdw = lds + g.off_wdw;
            const float* bdw = lds + g.off_bdw;
            f32x16c D[MT];
            auto dw = [&](int j, float4& bf) {
                const int c0 = h * Chp + 4 * j;
                bf = cld4(bdw + c0);
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        // SAME on an even size: no pad before, one zero row / column after.  The load is unconditional (clamped
                        // address) and the tap weight is zeroed instead: a predicated load would put a branch and a full
                        // memory round trip around every one of the nine taps
                        const int iy = 2 * oy + ky, ix = 2 * ox + kx;
                        const bool inside = iy < Hi && ix < Wi;
                        float4 w = cld4(wdw + (ky * 3 + kx) * g.Cp + c0);
                        if (!inside) w = zero4;
                        const float4 d = cld4(src + ((long)min(iy, Hi - 1) * Wi + min(ix, Wi - 1)) * Cin + c0);
                        bf.x = fmaf(d.x, w.x, bf.x);
                        bf.y = fmaf(d.y, w.y, bf.y);
                        bf.z = fmaf(d.z, w.z, bf.z);
                        bf.w = fmaf(d.w, w.w, bf.w);
                    }
            };
            contract(cb, Chp >> 2, dw, D);
            const float hi = cb.act == ACT_RELU6 ? 6.f : INFINITY;
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int ch = m * 32 + 8 * gq + 4 * h;
                    if (ch >= a.C) continue;
                    float4 sk = zero4;
                    if (cb.has_res && ch < Cin) {  // 2x2 max-pool of the input, channels above Cin are the zero pad
                        const float* p0 = src + ((long)(2 * oy) * Wi + 2 * ox) * Cin + ch;
                        const float4 s0 = cld4(p0), s1 = cld4(p0 + Cin), s2 = cld4(p0 + (long)Wi * Cin), s3 = cld4(p0 + (long)Wi * Cin + Cin);
                        sk = make_float4(fmaxf(fmaxf(s0.x, s1.x), fmaxf(s2.x, s3.x)), fmaxf(fmaxf(s0.y, s1.y), fmaxf(s2.y, s3.y)),
                                         fmaxf(fmaxf(s0.z, s1.z), fmaxf(s2.z, s3.z)), fmaxf(fmaxf(s0.w, s1.w), fmaxf(s2.w, s3.w)));
                    }
                    finish(D, m, gq, sk, hi);
                    if (valid) *reinterpret_cast<float4*>(centre + ch) = make_float4(D[m][4 * gq], D[m][4 * gq + 1], D[m][4 * gq + 2], D[m][4 * gq + 3]);
                }
        }
        __syncthreads();
    } else {
        // ---- load the frame (coalesced)
        __syncthreads();
        for (int i = tid; i < a.H * rowf4; i += 512) {
            int r = i / rowf4, e = i - r * rowf4;
            int px = e / g.C4, c4 = e - px * g.C4;
            *reinterpret_cast<float4*>(tile + (r + 1) * g.RS + (px + 1) * g.PS + 4 * c4) = cld4(in + 4 * (long)i);
        }
    }

    for (int blk = 0; blk < a.nblocks; blk++) {
        const ChainBlock& cb = a.blocks[blk];
        // ---- stage this block's small constants (previous block's readers are past the barrier below)
        stage_consts(cb, a.C, a.C);
        __syncthreads();
        const float* wdw = lds + g.off_wdw;
        const float* bdw = lds + g.off_bdw;

        f32x16c D[MT];
        if (wave_active) {
            auto dw = [&](int j, float4& bf) {
                const float* wj = wdw + h * g.Ch + 4 * j;
                bf = zero4;
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
            contract(cb, nch, dw, D);
            // ---- epilogue into registers (reads x at the centre pixel), written back after the barrier
            const float hi = cb.act == ACT_RELU6 ? 6.f : INFINITY;
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int ch = m * 32 + 8 * gq + 4 * h;
                    if (ch >= a.C) continue;
                    finish(D, m, gq, cb.has_res ? cld4(centre + ch) : zero4, hi);
                }
        }
        __syncthreads();  // every wave has read x for this block
        if (wave_active && valid) {
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int ch = m * 32 + 8 * gq + 4 * h;
                    if (ch >= a.C) continue;
                    *reinterpret_cast<float4*>(centre + ch) = make_float4(D[m][4 * gq], D[m][4 * gq + 1], D[m][4 * gq + 2], D[m][4 * gq + 3]);
                }
        }
        __syncthreads();
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
    if (a.post.on) {
        // ---- the stride-2 block behind the chain: taps from the resident frame, output straight to global memory
        const ChainBlock& cb = a.post.blk;
        const int Ho = a.H >> 1, Wo = a.W >> 1, Co = a.post.Co, npo = Ho * Wo;
        stage_consts(cb, a.C, Co);
        __syncthreads();
        const float* wdw = lds + g.off_wdw;
        const float* bdw = lds + g.off_bdw;
        for (int grp = wave; grp * 32 < npo; grp += 8) {  // wave-uniform
            const int qo = grp * 32 + pl;
            const bool vo = qo < npo;
            const int py = vo ? qo / Wo : 0, px = vo ? qo - (qo / Wo) * Wo : 0;
            // SAME on an even size: taps at image rows 2py .. 2py+2 = tile slots 2py+1 .. 2py+3 (slot H+1 is the zero border)
            const float* t0 = tile + (2 * py + 1) * g.RS + (2 * px + 1) * g.PS;
            f32x16c D[MT];
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
            contract(cb, nch, dw, D);
            const float hi = cb.act == ACT_RELU6 ? 6.f : INFINITY;
            float* dst = a.post.out + (long)b * a.post.out_fs + (long)qo * Co;
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int ch = m * 32 + 8 * gq + 4 * h;
                    if (ch >= Co) continue;
                    float4 sk = zero4;
                    if (cb.has_res && ch < a.C) {  // 2x2 max-pool of the resident frame, zero channel pad above C
                        const float4 s0 = cld4(t0 + ch), s1 = cld4(t0 + g.PS + ch), s2 = cld4(t0 + g.RS + ch), s3 = cld4(t0 + g.RS + g.PS + ch);
                        sk = make_float4(fmaxf(fmaxf(s0.x, s1.x), fmaxf(s2.x, s3.x)), fmaxf(fmaxf(s0.y, s1.y), fmaxf(s2.y, s3.y)),
                                         fmaxf(fmaxf(s0.z, s1.z), fmaxf(s2.z, s3.z)), fmaxf(fmaxf(s0.w, s1.w), fmaxf(s2.w, s3.w)));
                    }
                    finish(D, m, gq, sk, hi);
                    if (vo) {
                        const float4 v = make_float4(D[m][4 * gq], D[m][4 * gq + 1], D[m][4 * gq + 2], D[m][4 * gq + 3]);
                        *reinterpret_cast<float4*>(dst + ch) = v;
                        if (g.off_t8 >= 0) *reinterpret_cast<float4*>(lds + g.off_t8 + qo * (Co + 4) + ch) = v;
                    }
                }
        }
    }
    // ---- output heads: stacked 1x1 convolutions on the resident frame / on `post`'s output; unit = (32 pixels, 32 stacked rows)
    for (int hd = 0; hd < kChainHeads; hd++) {
        const ChainHead& H = a.heads[hd];
        if (!H.on) continue;
        __syncthreads();  // `post`'s LDS copy is complete (and every earlier stage is done with the constants)
        const bool from_post = H.src == 1;
        const int Wh = from_post ? a.W >> 1 : a.W, np = from_post ? (a.H >> 1) * (a.W >> 1) : a.H * a.W;
        const int Cs = from_post ? a.post.Co : a.C, Chs = Cs >> 1, nchs = Chs >> 2;
        const int Cot = H.Co_a + H.Co_b, MTh = (Cot + 31) >> 5, ng = (np + 31) >> 5;
        for (int u = wave; u < ng * MTh; u += 8) {  // wave-uniform
            const int grp = u % ng, mt = u / ng;
            const int qh = grp * 32 + pl;
            const bool vh = qh < np;
            const int qc = vh ? qh : 0;
            const float* px = from_post ? lds + g.off_t8 + qc * (Cs + 4) : tile + (qc / Wh + 1) * g.RS + (qc % Wh + 1) * g.PS;
            f32x16c D1;
#pragma unroll
            for (int e = 0; e < 16; e++) D1[e] = 0.f;
            const float* wa = H.w_pw + ((long)mt * nchs * 64 + lane) * 4;
            float4 av = cld4(wa);
            for (int j = 0; j < nchs; j++) {
                const float4 an = cld4(wa + 256 * min(j + 1, nchs - 1));  // next chunk's weights under this chunk's MFMAs
                const float4 bf = cld4(px + h * Chs + 4 * j);
                D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf.x, D1, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf.y, D1, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf.z, D1, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf.w, D1, 0, 0, 0);
                av = an;
            }
            if (!vh) continue;
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const int ch = mt * 32 + 8 * gq + 4 * h;   // stacked row
                if (ch >= Cot) continue;
                float v[4] = {D1[4 * gq], D1[4 * gq + 1], D1[4 * gq + 2], D1[4 * gq + 3]};
                if (H.bias) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] += ch + e < Cot ? H.bias[ch + e] : 0.f;
                }
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
}

bool make_chain_geom(const ChainArgs& a, ChainGeom* out) {
    ChainGeom g{};
    if (a.C % 8 || a.H * a.W > 256 || a.nblocks < 1 || a.nblocks > kMaxChain) return false;
    g.Cp = a.C; g.Ch = a.C / 2; g.C4 = a.C / 4; g.PS = a.C + 4; g.RS = (a.W + 2) * g.PS;
    g.MT = (a.C + 31) / 32;
    if (g.MT > 4) return false;
    int off = (a.H + 2) * g.RS;
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

template <int MT>
int launch_chain_inst(const ChainArgs& a, const ChainGeom& g, hipStream_t s) {
    auto kern = chain_kernel<MT>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
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
    hipStream_t s = (hipStream_t)stream;
    switch (g.MT) {
        case 1: return launch_chain_inst<1>(a, g, s);
        case 2: return launch_chain_inst<2>(a, g, s);
        case 3: return launch_chain_inst<3>(a, g, s);
        case 4: return launch_chain_inst<4>(a, g, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace mi
