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
// Mapping: a stage is cut into units of 32 output pixels x MTG 32-channel tiles; the 8 waves take units round-robin.  A unit
// is the block kernel's inner loop: lane = (pixel l & 31, k-half l >> 5) builds the B operand of v_mfma_f32_32x32x2_f32 for 4
// channels of its k-half (depthwise 3x3 on the VALU from LDS, or a 16-byte gather), the pointwise weights arrive from L2 in
// A-fragment order through a register ring that runs 8 k-chunks ahead (a stage is a chain of dependent MFMAs fed by L2
// latency; nothing else hides it here), D gives each lane 4 consecutive output channels per register quad for the float4
// epilogue (+ bias + skip [direct | 2x2 max-pool, zero channel-padded] -> activation -> LDS and/or global).
// Exact f32 (MFMA f32 = fmaf chain); results match block_kernels.hip to reassociation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "kernels.hpp"

namespace mi {

namespace {

typedef float rf32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 rld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 rmax4(float4 a, float4 b, float4 c, float4 d) {
    return make_float4(fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y)),
                       fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)), fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w)));
}
__device__ __forceinline__ float* resolve(const ResBases& bs, const ResRef& r, int frame) {
    return bs.p[r.base] + r.root_off * bs.scale[r.base] + r.inner + (long)(frame + bs.frame0[r.base]) * r.fs;
}

constexpr int kPFtot = 8;  // A fragments (float4) in flight per lane: 8 k-chunks ahead with one tile per unit, 4 with two

// MODE 0: GATHER from LDS, 1: GATHER from global memory, 2: DW from LDS
template <int MTG, int MODE>
__device__ __forceinline__ void run_units(const ResStage& st, const ResBases& bs, int frame, float* lds, const float* cst) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int npix = st.Ho * st.Wo, PGn = (npix + 31) >> 5, MT = (st.Co + 31) >> 5, MGn = (MT + MTG - 1) / MTG;
    const int nunits = PGn * MGn;
    const int Cp = (st.Kv + 7) & ~7, Ch = Cp >> 1, nch = Ch >> 2;
    const int Cop = MT * 32;
    const float* wdw = cst;                    // [9][Cp] (DW only)
    const float* bdw = cst + 9 * Cp;           // [Cp]
    const float* biasL = cst + (MODE == 2 ? 10 * Cp : 0);
    const float* alphaL = biasL + Cop;
    const float* wpw = bs.weights + st.w_pw;
    const float* srcg = MODE == 1 ? resolve(bs, st.src_g, frame) : nullptr;
    const float* resg = (st.res_mode != RES_NONE && st.res_off < 0) ? resolve(bs, st.res_g, frame) : nullptr;
    float* dstg = st.dst_g.base >= 0 ? resolve(bs, st.dst_g, frame) : nullptr;
    const float hi = st.act == ACT_RELU6 ? 6.f : INFINITY;
    const int srow = st.src_W + 2 * st.src_b;  // LDS row length of the source in pixels

    for (int u = wave; u < nunits; u += 8) {
        const int pg = u % PGn, mt0 = (u / PGn) * MTG;
        const int q = pg * 32 + pl;
        const bool valid = q < npix;
        const int qq = valid ? q : 0;
        const int oy = qq / st.Wo, ox = qq - oy * st.Wo;

        rf32x16 D[MTG];
#pragma unroll
        for (int m = 0; m < MTG; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) D[m][e] = 0.f;

        auto a_load = [&](int j, float4 (&av)[MTG]) {
#pragma unroll
            for (int m = 0; m < MTG; m++) {
                const int mt = min(mt0 + m, MT - 1);
                av[m] = rld4(wpw + (((long)mt * nch + j) * 64 + lane) * 4);
            }
        };
        // B fragment of k-chunk j: 4 consecutive (virtual) channels of this lane's k-half at its pixel
        auto gather_addr = [&](int j, bool& ok) -> long {
            const int v0 = h * Ch + 4 * j;
            ok = v0 < st.Kv;
            const int vv = ok ? v0 : 0;
            const int tap = vv / st.src_C, c = vv - tap * st.src_C;
            const int ky = tap / st.KW, kx = tap - ky * st.KW;
            const int iy = oy * st.S + ky, ix = ox * st.S + kx;
            if (MODE == 1) return ((long)iy * st.src_W + ix) * st.src_C + c;
            return (long)st.src_off + ((iy + st.src_b) * srow + ix + st.src_b) * st.src_PS + c;
        };
        auto b_lds = [&](int j) -> float4 {
            if (MODE == 0) {
                bool ok;
                const long a = gather_addr(j, ok);
                const float4 v = rld4(lds + a);
                return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // depthwise 3x3: channels h*Ch + 4j .. +3 (weights and the LDS pad lanes are zero above the real channel count)
            const int c0 = h * Ch + 4 * j;
            const float* p = lds + st.src_off + ((oy * st.S - st.pt + st.src_b) * srow + ox * st.S - st.pl + st.src_b) * st.src_PS + c0;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const float4 w = rld4(wdw + (ky * 3 + kx) * Cp + c0);
                    const float4 d = rld4(p + (ky * srow + kx) * st.src_PS);
                    acc.x = fmaf(d.x, w.x, acc.x);
                    acc.y = fmaf(d.y, w.y, acc.y);
                    acc.z = fmaf(d.z, w.z, acc.z);
                    acc.w = fmaf(d.w, w.w, acc.w);
                }
            const float4 bb = rld4(bdw + c0);
            acc.x += bb.x; acc.y += bb.y; acc.z += bb.z; acc.w += bb.w;
            return acc;
        };
        auto b_glb = [&](int j) -> float4 {
            bool ok;
            const long a = gather_addr(j, ok);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = rld4(srcg + a);
            return v;
        };
        auto mfma4 = [&](const float4 (&av)[MTG], const float4& bf) {
#pragma unroll
            for (int m = 0; m < MTG; m++) {
                D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].x, bf.x, D[m], 0, 0, 0);
                D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].y, bf.y, D[m], 0, 0, 0);
                D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].z, bf.z, D[m], 0, 0, 0);
                D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].w, bf.w, D[m], 0, 0, 0);
            }
        };

        constexpr int kPF = (MODE == 1 ? kPFtot / 2 : kPFtot) / MTG;
        float4 ringA[kPF][MTG];
        float4 ringB[MODE == 1 ? kPF : 1];
#pragma unroll
        for (int r = 0; r < kPF; r++)
            if (r < nch) {
                a_load(r, ringA[r]);
                if (MODE == 1) ringB[r] = b_glb(r);
            }
        for (int j0 = 0; j0 < nch; j0 += kPF) {
#pragma unroll
            for (int r = 0; r < kPF; r++) {
                const int j = j0 + r;
                if (j < nch) {
                    const float4 bf = MODE == 1 ? ringB[MODE == 1 ? r : 0] : b_lds(j);
                    mfma4(ringA[r], bf);
                    if (j + kPF < nch) {
                        a_load(j + kPF, ringA[r]);
                        if (MODE == 1) ringB[MODE == 1 ? r : 0] = b_glb(j + kPF);
                    }
                }
            }
        }

        // ---- epilogue: lane holds pixel q, output channels mt*32 + 8*gq + 4*h .. +3 in D[m][4*gq .. 4*gq+3]
        if (valid) {
            float* dl = st.dst_off >= 0 ? lds + st.dst_off + ((oy + st.dst_b) * (st.Wo + 2 * st.dst_b) + ox + st.dst_b) * st.dst_PS : nullptr;
            float* dg = dstg ? dstg + ((long)oy * st.Wo + ox) * st.Co : nullptr;
#pragma unroll
            for (int m = 0; m < MTG; m++) {
                if (mt0 + m >= MT) continue;
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int ch = (mt0 + m) * 32 + 8 * gq + 4 * h;
                    if (ch >= st.Co) continue;
                    const float4 bb = rld4(biasL + ch), al = rld4(alphaL + ch);
                    float4 v = make_float4(D[m][4 * gq] + bb.x, D[m][4 * gq + 1] + bb.y, D[m][4 * gq + 2] + bb.z, D[m][4 * gq + 3] + bb.w);
                    if (st.res_mode != RES_NONE && ch < st.res_C) {
                        float4 rv;
                        if (st.res_off >= 0) {
                            const int rrow = st.res_W + 2 * st.res_b;
                            if (st.res_mode == RES_DIRECT) {
                                rv = rld4(lds + st.res_off + ((oy + st.res_b) * rrow + ox + st.res_b) * st.res_PS + ch);
                            } else {
                                const float* p = lds + st.res_off + ((2 * oy + st.res_b) * rrow + 2 * ox + st.res_b) * st.res_PS + ch;
                                rv = rmax4(rld4(p), rld4(p + st.res_PS), rld4(p + rrow * st.res_PS), rld4(p + (rrow + 1) * st.res_PS));
                            }
                        } else {
                            if (st.res_mode == RES_DIRECT) {
                                rv = rld4(resg + ((long)oy * st.Wo + ox) * st.res_C + ch);
                            } else {
                                const float* p = resg + ((long)(2 * oy) * st.res_W + 2 * ox) * st.res_C + ch;
                                rv = rmax4(rld4(p), rld4(p + st.res_C), rld4(p + (long)st.res_W * st.res_C), rld4(p + (long)(st.res_W + 1) * st.res_C));
                            }
                        }
                        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                    }
                    v.x = fminf(fmaxf(v.x, 0.f) + al.x * fminf(v.x, 0.f), hi);
                    v.y = fminf(fmaxf(v.y, 0.f) + al.y * fminf(v.y, 0.f), hi);
                    v.z = fminf(fmaxf(v.z, 0.f) + al.z * fminf(v.z, 0.f), hi);
                    v.w = fminf(fmaxf(v.w, 0.f) + al.w * fminf(v.w, 0.f), hi);
                    if (dl) *reinterpret_cast<float4*>(dl + ch) = v;   // Co % 4 == 0 whenever the output stays in LDS (planner)
                    if (dg) {
                        if ((st.Co & 3) == 0) {
                            *reinterpret_cast<float4*>(dg + ch) = v;
                        } else {  // ragged output heads (213, 15, 1 channels)
                            dg[ch] = v.x;
                            if (ch + 1 < st.Co) dg[ch + 1] = v.y;
                            if (ch + 2 < st.Co) dg[ch + 2] = v.z;
                            if (ch + 3 < st.Co) dg[ch + 3] = v.w;
                        }
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(512, 4) void resident_kernel(const ResStage* __restrict__ prog, int nstages, int const_off, ResBases bs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int frame = blockIdx.x;
    float* cst = lds + const_off;
    for (int s = 0; s < nstages; s++) {
        const ResStage& st = prog[s];
        // ---- phase 0: clear the destination (zero border / zero pad lanes), stage the small constants
        for (int i = tid; i < (st.zero_dst >> 2); i += 512) reinterpret_cast<float4*>(lds + st.dst_off)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (st.kind != RES_STAGE_LOAD) {
            const int Cp = (st.Kv + 7) & ~7, Cop = ((st.Co + 31) >> 5) * 32;
            int o = 0;
            if (st.kind == RES_STAGE_DW) {
                const float* w = bs.weights + st.w_dw;
                for (int i = tid; i < 9 * Cp; i += 512) {
                    const int c = i % Cp;
                    cst[i] = c < st.Kv ? w[(i / Cp) * st.Kv + c] : 0.f;
                }
                for (int i = tid; i < Cp; i += 512) cst[9 * Cp + i] = (i < st.Kv && st.b_dw >= 0) ? bs.weights[st.b_dw + i] : 0.f;
                o = 10 * Cp;
            }
            for (int i = tid; i < Cop; i += 512) {
                cst[o + i] = (i < st.Co && st.bias >= 0) ? bs.weights[st.bias + i] : 0.f;
                cst[o + Cop + i] = (i < st.Co && st.act == ACT_PRELU) ? bs.weights[st.alpha + i] : (st.act == ACT_NONE ? 1.f : 0.f);
            }
        }
        __syncthreads();
        // ---- phase 1
        if (st.kind == RES_STAGE_LOAD) {
            const float* src = resolve(bs, st.src_g, frame);
            const int C4 = st.src_C >> 2, n = st.src_H * st.src_W * C4;
            const int drow = st.src_W + 2 * st.dst_b;
            for (int i = tid; i < n; i += 512) {
                const int px = i / C4, c4 = i - px * C4;
                const int y = px / st.src_W, x = px - y * st.src_W;
                *reinterpret_cast<float4*>(lds + st.dst_off + ((y + st.dst_b) * drow + x + st.dst_b) * st.dst_PS + 4 * c4) = rld4(src + 4 * (long)i);
            }
        } else if (st.kind == RES_STAGE_DW) {
            run_units<1, 2>(st, bs, frame, lds, cst);
        } else if (st.src_off >= 0) {
            run_units<1, 0>(st, bs, frame, lds, cst);
        } else {
            run_units<1, 1>(st, bs, frame, lds, cst);
        }
        __syncthreads();
    }
}

}  // namespace

int resident_const_floats(const ResStage& st) {
    if (st.kind == RES_STAGE_LOAD) return 0;
    const int Cp = (st.Kv + 7) & ~7, Cop = (st.Co + 31) / 32 * 32;
    return (st.kind == RES_STAGE_DW ? 10 * Cp : 0) + 2 * Cop;
}

int launch_resident(const ResLaunch& a, void* stream) {
    if (!a.prog || a.nstages < 1 || a.B < 1 || a.lds_bytes > 160 * 1024) return (int)hipErrorInvalidValue;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    hipLaunchKernelGGL(resident_kernel, dim3((unsigned)a.B), dim3(512), (size_t)a.lds_bytes, (hipStream_t)stream, a.prog, a.nstages, a.const_off, a.bases);
    return (int)hipGetLastError();
}

}  // namespace mi
