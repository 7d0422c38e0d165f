// block_kernels.hip — the hot kernel: one fused BlazeBlock per launch on gfx950 (CDNA4).
//
//   out = act( PW1x1( DW3x3(in) + b_dw ) + b_pw + skip )            (HAS_DW)      or
//   out = act( PW1x1( in ) + b_pw + skip )                           (pointwise only)
//
// replaces the DEPTHWISE_CONV_2D -> CONV_2D(1x1) -> ADD -> RELU/PRELU op chain (+ MAX_POOL_2D / channel PAD on the skip
// path) that TensorFlow-Lite runs op by op behind `interpreter.invoke()` (/root/reference/src/face_detection_lite/
// face_detection.rs:235, face_landmark.rs:265, iris_landmark.rs:203; graphs: SURVEY.md Appendix A).
//
// Design (HBM-bound: every activation byte is read once and written once):
//   * one 256-thread workgroup owns a band of output rows of ONE frame and walks down it.  Input rows live in an LDS
//     ring (NR = (R-1)*S + 3 rows for R output rows per step); each input row is fetched from HBM exactly once per band
//     (2 halo rows per band are the only re-reads).  Rows of an NHWC frame are contiguous, so the 16 B/lane loads and
//     stores are fully coalesced.
//   * the next step's rows are issued into registers BEFORE the current step is computed and written to LDS after it
//     (issue-early / write-late), so HBM latency hides under the DW + MFMA work; 2-3 workgroups per CU cover the rest.
//   * LDS pixel stride is Cp + 4 floats: with Cp a multiple of 8 the 16 lanes of a ds_read_b128 group hit 16 distinct
//     4-bank slots (conflict-free) when each lane reads its own pixel.
//   * depthwise 3x3 on the VALU, 4 channels (one float4) at a time, computed directly in the MFMA operand layout:
//     lane l = (pixel l&31, k-half l>>5) owns the channels [h*Cp/2, (h+1)*Cp/2) of its pixel, so the DW result IS the
//     B operand of v_mfma_f32_32x32x2_f32 (B[k = l>>5][n = l&31]) with the contraction index permuted consistently in
//     the pre-packed A operand (pointwise weights, M = 32 output channels per tile).  No LDS round trip for DW outputs.
//   * exact f32 MFMA (same numerics as an fmaf chain); D layout gives each lane 4 consecutive output channels per
//     register quad -> float4 epilogue: + bias + skip (from the LDS ring when the skip is the block input, incl. the
//     fused 2x2 max-pool of stride-2 blocks and the zero channel-pad) -> ReLU/PReLU -> 16-byte stores.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "kernels.hpp"

namespace mi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kPrefetch = 8;        // float4 registers per thread for the next step's rows
constexpr int kLdsBudget = 64 * 1024;
constexpr int kALdsMax = 40 * 1024; // pointwise weights are staged in LDS when they fit in this many bytes

struct BlockGeom {
    int Cp, Ch, C4;        // padded channels (mult of 8), channels per k-half, float4s per real pixel
    int PS, RS;            // LDS pixel stride / row stride in floats
    int R, NR;             // output rows per step, ring rows
    int band, bands;       // output rows per workgroup, workgroups per frame
    int MT;                // 32-row output-channel tiles
    int a_lds;             // pointwise weights staged in LDS
    int off_wdw, off_bdw, off_a;  // LDS offsets in floats
    int res_lds;           // skip connection served from the LDS ring
    int lds_bytes;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ float4 res_from_global(const Epilogue& ep, int b, int oy, int ox, int Wo, int ch) {
    // generic float4 skip read (channels ch..ch+3 < res_C guaranteed by the caller)
    const float* r = ep.res + (long)b * ep.res_fs;
    if (ep.res_mode == RES_DIRECT) return ld4(r + ((long)oy * Wo + ox) * ep.res_C + ch);
    if (ep.res_mode == RES_MAXPOOL) {
        const float* p = r + ((long)(2 * oy) * ep.res_W + 2 * ox) * ep.res_C + ch;
        float4 a = ld4(p), b4 = ld4(p + ep.res_C), c = ld4(p + (long)ep.res_W * ep.res_C), d = ld4(p + (long)ep.res_W * ep.res_C + ep.res_C);
        return make_float4(fmaxf(fmaxf(a.x, b4.x), fmaxf(c.x, d.x)), fmaxf(fmaxf(a.y, b4.y), fmaxf(c.y, d.y)),
                           fmaxf(fmaxf(a.z, b4.z), fmaxf(c.z, d.z)), fmaxf(fmaxf(a.w, b4.w), fmaxf(c.w, d.w)));
    }
    // RES_UP2X (TFLite ResizeBilinear, half_pixel_centers)
    float iy = ((float)oy + 0.5f) * 0.5f - 0.5f, ix = ((float)ox + 0.5f) * 0.5f - 0.5f;
    int y0 = max((int)floorf(iy), 0), y1 = min((int)ceilf(iy), ep.res_H - 1);
    int x0 = max((int)floorf(ix), 0), x1 = min((int)ceilf(ix), ep.res_W - 1);
    float dy = iy - (float)y0, dx = ix - (float)x0;
    float4 p00 = ld4(r + ((long)y0 * ep.res_W + x0) * ep.res_C + ch), p01 = ld4(r + ((long)y0 * ep.res_W + x1) * ep.res_C + ch);
    float4 p10 = ld4(r + ((long)y1 * ep.res_W + x0) * ep.res_C + ch), p11 = ld4(r + ((long)y1 * ep.res_W + x1) * ep.res_C + ch);
    float w00 = (1 - dy) * (1 - dx), w10 = dy * (1 - dx), w01 = (1 - dy) * dx, w11 = dy * dx;
    return make_float4(p00.x * w00 + p10.x * w10 + p01.x * w01 + p11.x * w11, p00.y * w00 + p10.y * w10 + p01.y * w01 + p11.y * w11,
                       p00.z * w00 + p10.z * w10 + p01.z * w01 + p11.z * w11, p00.w * w00 + p10.w * w10 + p01.w * w01 + p11.w * w11);
}

// MTG: output-channel tiles kept in accumulators at once; S: stride; KS: 3 (depthwise stage) or 1 (pointwise only);
// PG: 32-pixel groups per wave per step.
template <int MTG, int S, int KS, int PG>
__global__ __launch_bounds__(256) void block_kernel(BlockArgs a, BlockGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tile = lds;
    const float* wdw = lds + g.off_wdw;
    const float* bdw = lds + g.off_bdw;
    const float* aL = lds + g.off_a;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / g.bands, band = blockIdx.x % g.bands;
    const int oy0 = band * g.band, oy1 = min(oy0 + g.band, a.Ho);
    const int iy_base = oy0 * S - a.pt;  // input row held by ring slot 0 at band start
    const float* in = a.in + (long)b * a.in_fs;
    const int rowf4 = a.W * g.C4;        // float4s of one real input row

    // ---- prologue: clear the ring (pad pixels / out-of-image rows stay zero), stage weights, load the first rows
    for (int i = tid; i < (g.NR * g.RS) >> 2; i += 256) reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KS == 3) {
        for (int i = tid; i < 9 * g.Cp; i += 256) {
            int c = i % g.Cp;
            lds[g.off_wdw + i] = c < a.C ? a.w_dw[(i / g.Cp) * a.C + c] : 0.f;
        }
        for (int i = tid; i < g.Cp; i += 256) lds[g.off_bdw + i] = (i < a.C && a.b_dw) ? a.b_dw[i] : 0.f;
    }
    if (g.a_lds)
        for (int i = tid; i < (g.MT * 32 * g.Cp) >> 2; i += 256) reinterpret_cast<float4*>(lds + g.off_a)[i] = ld4(a.w_pw + 4 * (long)i);
    __syncthreads();
    {
        const int nrows = min(g.NR, (oy1 - oy0 - 1) * S + KS);
        for (int i = tid; i < nrows * rowf4; i += 256) {
            int r = i / rowf4, e = i - r * rowf4;
            int px = e / g.C4, c4 = e - px * g.C4;
            int iy = iy_base + r;
            if (iy >= 0 && iy < a.H)
                *reinterpret_cast<float4*>(tile + r * g.RS + (px + 1) * g.PS + 4 * c4) = ld4(in + ((long)iy * a.W + px) * a.C + 4 * c4);
        }
    }
    __syncthreads();

    const int nsteps = (oy1 - oy0 + g.R - 1) / g.R;
    for (int step = 0; step < nsteps; step++) {
        const int oys = oy0 + step * g.R;                  // first output row of this step
        // ---- issue the loads of the rows the NEXT step adds to the ring (kept in registers across the compute phase)
        float4 pf[kPrefetch];
        const int new_lo = (oys - oy0 + g.R - 1) * S + KS; // ring-relative index of the first new row
        const bool more = step + 1 < nsteps;
        const int new_n = more ? min(g.R, oy1 - (oys + g.R)) * S : 0;  // rows the next step really needs
        if (more) {
#pragma unroll
            for (int k = 0; k < kPrefetch; k++) {
                int i = tid + k * 256;
                pf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < new_n * rowf4) {
                    int r = i / rowf4, e = i - r * rowf4;
                    int px = e / g.C4, c4 = e - px * g.C4;
                    int iy = iy_base + new_lo + r;
                    if (iy >= 0 && iy < a.H) pf[k] = ld4(in + ((long)iy * a.W + px) * a.C + 4 * c4);
                }
            }
        }

        // ---- compute: this wave's PG groups of 32 output pixels
        int q[PG], oy[PG], ox[PG];
        bool valid[PG];
        int base[PG][KS];  // LDS float offset of tap row ky, kx = 0, channel h*Ch
#pragma unroll
        for (int p = 0; p < PG; p++) {
            q[p] = (wave * PG + p) * 32 + pl;
            int r = q[p] / a.Wo;
            oy[p] = oys + r;
            ox[p] = q[p] - r * a.Wo;
            valid[p] = r < g.R && oy[p] < oy1;
            int oyc = valid[p] ? oy[p] : oys, oxc = valid[p] ? ox[p] : 0;
#pragma unroll
            for (int ky = 0; ky < KS; ky++) {
                int rel = (oyc - oy0) * S + ky;
                base[p][ky] = (rel % g.NR) * g.RS + (oxc * S - a.pl + 1) * g.PS + h * g.Ch;
            }
        }
        for (int mt0 = 0; mt0 < g.MT; mt0 += MTG) {
            f32x16 D[PG][MTG];
#pragma unroll
            for (int p = 0; p < PG; p++)
#pragma unroll
                for (int m = 0; m < MTG; m++)
#pragma unroll
                    for (int e = 0; e < 16; e++) D[p][m][e] = 0.f;
            for (int j = 0; j < (g.Ch >> 2); j++) {
                float4 bf[PG];
                if (KS == 3) {
#pragma unroll
                    for (int p = 0; p < PG; p++) bf[p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            float4 w = ld4(wdw + (ky * 3 + kx) * g.Cp + h * g.Ch + 4 * j);
#pragma unroll
                            for (int p = 0; p < PG; p++) {
                                float4 d = ld4(tile + base[p][ky] + kx * g.PS + 4 * j);
                                bf[p].x = fmaf(d.x, w.x, bf[p].x);
                                bf[p].y = fmaf(d.y, w.y, bf[p].y);
                                bf[p].z = fmaf(d.z, w.z, bf[p].z);
                                bf[p].w = fmaf(d.w, w.w, bf[p].w);
                            }
                        }
                    float4 bb = ld4(bdw + h * g.Ch + 4 * j);
#pragma unroll
                    for (int p = 0; p < PG; p++) { bf[p].x += bb.x; bf[p].y += bb.y; bf[p].z += bb.z; bf[p].w += bb.w; }
                } else {
#pragma unroll
                    for (int p = 0; p < PG; p++) bf[p] = ld4(tile + base[p][0] + 4 * j);
                }
#pragma unroll
                for (int m = 0; m < MTG; m++) {
                    if (mt0 + m < g.MT) {
                        // A fragment of tile (mt0+m), k-steps 4j..4j+3: packed [mt][j][lane][4]
                        long ao = (((long)(mt0 + m) * (g.Ch >> 2) + j) * 64 + lane) * 4;
                        float4 av = g.a_lds ? ld4(aL + ao) : ld4(a.w_pw + ao);
#pragma unroll
                        for (int p = 0; p < PG; p++) {
                            D[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf[p].x, D[p][m], 0, 0, 0);
                            D[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf[p].y, D[p][m], 0, 0, 0);
                            D[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf[p].z, D[p][m], 0, 0, 0);
                            D[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf[p].w, D[p][m], 0, 0, 0);
                        }
                    }
                }
            }
            // ---- epilogue: lane holds pixel pl, output channels (mt*32 + 8*gq + 4*h .. +3) in D[..][4*gq .. 4*gq+3]
#pragma unroll
            for (int p = 0; p < PG; p++) {
                if (!valid[p]) continue;
                float* op = a.out + (long)b * a.out_fs + ((long)oy[p] * a.Wo + ox[p]) * a.Co;
#pragma unroll
                for (int m = 0; m < MTG; m++) {
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        int ch = (mt0 + m) * 32 + 8 * gq + 4 * h;
                        if (mt0 + m >= g.MT || ch >= a.Co) continue;
                        float4 v = make_float4(D[p][m][4 * gq], D[p][m][4 * gq + 1], D[p][m][4 * gq + 2], D[p][m][4 * gq + 3]);
                        if (a.ep.bias) { float4 bb = ld4(a.ep.bias + ch); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
                        if (a.ep.res_mode != RES_NONE && ch < a.ep.res_C) {
                            float4 rv;
                            if (g.res_lds) {
                                if (a.ep.res_mode == RES_DIRECT) {
                                    int rel = (oy[p] - oy0) + a.pt;
                                    rv = ld4(tile + (rel % g.NR) * g.RS + (ox[p] + 1) * g.PS + ch);
                                } else {  // RES_MAXPOOL of the stride-2 block input (pt = pl = 0)
                                    int rel = (oy[p] - oy0) * 2;
                                    const float* r0 = tile + (rel % g.NR) * g.RS + (2 * ox[p] + 1) * g.PS + ch;
                                    const float* r1 = tile + ((rel + 1) % g.NR) * g.RS + (2 * ox[p] + 1) * g.PS + ch;
                                    float4 x0 = ld4(r0), x1 = ld4(r0 + g.PS), x2 = ld4(r1), x3 = ld4(r1 + g.PS);
                                    rv = make_float4(fmaxf(fmaxf(x0.x, x1.x), fmaxf(x2.x, x3.x)), fmaxf(fmaxf(x0.y, x1.y), fmaxf(x2.y, x3.y)),
                                                     fmaxf(fmaxf(x0.z, x1.z), fmaxf(x2.z, x3.z)), fmaxf(fmaxf(x0.w, x1.w), fmaxf(x2.w, x3.w)));
                                }
                            } else {
                                rv = res_from_global(a.ep, b, oy[p], ox[p], a.Wo, ch);
                            }
                            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                        }
                        if (a.ep.act == ACT_RELU) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        } else if (a.ep.act == ACT_PRELU) {
                            float4 al = ld4(a.ep.alpha + ch);
                            v.x = v.x >= 0.f ? v.x : al.x * v.x; v.y = v.y >= 0.f ? v.y : al.y * v.y;
                            v.z = v.z >= 0.f ? v.z : al.z * v.z; v.w = v.w >= 0.f ? v.w : al.w * v.w;
                        } else if (a.ep.act == ACT_RELU6) {
                            v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
                            v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
                        }
                        *reinterpret_cast<float4*>(op + ch) = v;
                    }
                }
            }
        }

        // ---- retire this step's dead rows: write the prefetched rows into their ring slots
        __syncthreads();
        if (more) {
#pragma unroll
            for (int k = 0; k < kPrefetch; k++) {
                int i = tid + k * 256;
                if (i < new_n * rowf4) {
                    int r = i / rowf4, e = i - r * rowf4;
                    int px = e / g.C4, c4 = e - px * g.C4;
                    *reinterpret_cast<float4*>(tile + ((new_lo + r) % g.NR) * g.RS + (px + 1) * g.PS + 4 * c4) = pf[k];
                }
            }
        }
        __syncthreads();
    }
}

bool make_geom(const BlockArgs& a, int PG, BlockGeom* out) {
    BlockGeom g{};
    const int S = a.sh, KS = a.has_dw ? 3 : 1;
    if (a.C % 4 || a.Co % 4 || a.sh != a.sw || (S != 1 && S != 2)) return false;
    if (!a.has_dw && S != 1) return false;
    g.Cp = (a.C + 7) & ~7;
    g.Ch = g.Cp / 2;
    g.C4 = a.C / 4;
    g.PS = g.Cp + 4;
    g.RS = (a.W + 2) * g.PS;
    g.MT = (a.Co + 31) / 32;
    const long a_bytes = (long)g.MT * 32 * g.Cp * 4;
    g.a_lds = a_bytes <= kALdsMax;
    const int fixed = (a.has_dw ? 10 * g.Cp * 4 : 0) + (g.a_lds ? (int)a_bytes : 0);
    auto lds_for = [&](int R) { return ((R - 1) * S + KS) * g.RS * 4 + fixed; };
    int R = std::max(1, std::min(a.Ho, (128 * PG) / a.Wo));
    while (R > 1 && (lds_for(R) > kLdsBudget || (long)R * S * a.W * g.C4 > (long)kPrefetch * 256)) R--;
    if (lds_for(R) > 150 * 1024 || (long)R * S * a.W * g.C4 > (long)kPrefetch * 256) return false;
    g.R = R;
    g.NR = (R - 1) * S + KS;
    const int max_bands = (a.Ho + R - 1) / R;
    int bands = std::min(max_bands, std::max(1, (1536 + a.B - 1) / std::max(1, a.B)));
    g.band = ((a.Ho + bands - 1) / bands + R - 1) / R * R;
    g.bands = (a.Ho + g.band - 1) / g.band;
    int off = g.NR * g.RS;
    g.off_wdw = off; off += a.has_dw ? 9 * g.Cp : 0;
    g.off_bdw = off; off += a.has_dw ? g.Cp : 0;
    off = (off + 3) & ~3;
    g.off_a = off; off += g.a_lds ? g.MT * 32 * g.Cp : 0;
    g.lds_bytes = off * 4;
    // skip connection straight from the ring when it is the block's own input tensor
    g.res_lds = 0;
    if (a.ep.res == a.in && a.ep.res_fs == a.in_fs && a.ep.res_C == a.C) {
        if (a.ep.res_mode == RES_DIRECT && S == 1) g.res_lds = 1;
        if (a.ep.res_mode == RES_MAXPOOL && S == 2 && a.has_dw && a.pt == 0 && a.pl == 0 && a.H == 2 * a.Ho && a.W == 2 * a.Wo) g.res_lds = 1;
    }
    if (a.ep.res_mode != RES_NONE && a.ep.res_C % 4) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!aligned16(a.in) || !aligned16(a.out) || (a.in_fs & 3) || (a.out_fs & 3)) return false;
    if (a.ep.res && (!aligned16(a.ep.res) || (a.ep.res_fs & 3))) return false;
    *out = g;
    return true;
}

template <int MTG, int S, int KS, int PG>
int launch_inst(const BlockArgs& a, const BlockGeom& g, hipStream_t s) {
    auto kern = block_kernel<MTG, S, KS, PG>;
    static bool configured = false;  // one attribute call per instantiation (process-wide; same for every device here)
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * g.bands)), dim3(256), (size_t)g.lds_bytes, s, a, g);
    return (int)hipGetLastError();
}

int pick_pg(const BlockArgs& a) {
    int MT = (a.Co + 31) / 32;
    return MT <= 2 ? 2 : 1;
}

}  // namespace

// Pointwise weights are packed on the host straight into MFMA A-fragment order (see pack_block_weights).
void block_weight_dims(int C, int Co, int* Cp, int* Cop) {
    *Cp = (C + 7) & ~7;
    *Cop = (Co + 31) / 32 * 32;
}

bool block_kernel_supports(const BlockArgs& a) {
    BlockGeom g;
    return make_geom(a, pick_pg(a), &g);
}

int launch_block(const BlockArgs& a, void* stream) {
    BlockGeom g;
    const int PG = pick_pg(a);
    if (!make_geom(a, PG, &g)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const int MTG = std::min(4, g.MT);
    const int S = a.sh;
#define MI_BLOCK_CASE(M, SS, K, P) \
    if (MTG == M && S == SS && (a.has_dw ? 3 : 1) == K && PG == P) return launch_inst<M, SS, K, P>(a, g, s);
    MI_BLOCK_CASE(1, 1, 3, 2) MI_BLOCK_CASE(2, 1, 3, 2) MI_BLOCK_CASE(3, 1, 3, 1) MI_BLOCK_CASE(4, 1, 3, 1)
    MI_BLOCK_CASE(1, 2, 3, 2) MI_BLOCK_CASE(2, 2, 3, 2) MI_BLOCK_CASE(3, 2, 3, 1) MI_BLOCK_CASE(4, 2, 3, 1)
    MI_BLOCK_CASE(1, 1, 1, 2) MI_BLOCK_CASE(2, 1, 1, 2) MI_BLOCK_CASE(3, 1, 1, 1) MI_BLOCK_CASE(4, 1, 1, 1)
#undef MI_BLOCK_CASE
    return (int)hipErrorInvalidValue;
}

}  // namespace mi
