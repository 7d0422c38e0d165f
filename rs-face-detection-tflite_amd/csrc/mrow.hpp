// mrow.hpp — device building blocks shared by the row-walking MFMA kernels (mdblock_kernels.hip, mwalk_kernels.hip): one input
// row of a depthwise 3x3 + pointwise stage in the v_mfma_f32_16x16x4_f32 operand layout.  Lane (kq = lane / 16, p = lane % 16) owns
// channel 4 ks + kq of pixel p of a 16-pixel tile for every k-step ks; see mstrip_kernels.hip for the scheme and its measurements.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

namespace mi {

namespace {

typedef float df32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void dwait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70); }
__device__ __forceinline__ void dwave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// workgroup barrier behind an LDS-only wait (__syncthreads() would also drain vmcnt: the next row's DMA and the output stores)
__device__ __forceinline__ void dwg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <class F, int... KS>
__device__ __forceinline__ void dfor_each(F&& f, std::integer_sequence<int, KS...>) { (f(std::integral_constant<int, KS>{}), ...); }

typedef float dv2f __attribute__((ext_vector_type(2)));

// Partial depthwise rows of a wave's WT pixel tiles for one k-step: tiles 0 / 1 (and 2 / 3) as register PAIRS, an odd last tile alone.
// A pair is one v_pk_fma_f32 per tap: the tap is a half of a register pair too, broadcast to both lanes of the packed operation by op_sel
// (a tap duplicated into both halves was what kept mstrip_kernel on scalar FMAs) — half the depthwise instructions of a row.
template <int WT>
struct RowAcc {
    dv2f p[WT / 2 ? WT / 2 : 1];
    float s[1];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int q = 0; q < (WT / 2 ? WT / 2 : 1); q++) p[q] = dv2f{0.f, 0.f};
        s[0] = 0.f;
    }
};

// acc += x * w.lo / w.hi on both lanes (S = 0: the low half of the tap pair, 1: the high half).  Inline asm: written as vector code
// (splat folded into op_sel by LLVM) the register allocator scalarises half of the operations again and spills.  The instructions are
// then invisible to the hazard recogniser (a v_mfma reads a register v_pk_fma_f32 wrote 0 or 1 instructions earlier STALE,
// tools/pk_mfma_hazard.hip): mrow_stage computes a pair's finished row first and puts a scheduling fence in front of the MFMAs, so
// that six packed operations lie between; tests/test_isa_checks.py keeps these kernels free of spills, which would break the count.
template <int S>
__device__ __forceinline__ void dpk_fma(dv2f& acc, const dv2f x, const dv2f w) {
    if constexpr (S == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(w));
    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(w));
}
template <int S>
__device__ __forceinline__ dv2f dpk_mul(const dv2f x, const dv2f w) {
    dv2f r;
    if constexpr (S == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(w));
    else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "v"(w));
    return r;
}

// The nine taps of a k-step as five register pairs: t[i] = (w[2 i], w[2 i + 1]); from three float4s [w0 w1 w2 w3][w4 w5 w6 w7][w8 . . .]
__device__ __forceinline__ void dtaps_from(const float4 t0, const float4 t1, const float4 t2, dv2f (&t)[5]) {
    t[0] = dv2f{t0.x, t0.y}; t[1] = dv2f{t0.z, t0.w}; t[2] = dv2f{t1.x, t1.y}; t[3] = dv2f{t1.z, t1.w}; t[4] = dv2f{t2.x, 0.f};
}
// tap[ks] <- this lane's taps (tp: its [ks][.][12] block in LDS), resident in registers for the whole kernel
template <int CK>
__device__ __forceinline__ void dload_taps(const float4* tp, dv2f (&tap)[CK][5]) {
#pragma unroll
    for (int ks = 0; ks < CK; ks++) dtaps_from(tp[ks * 12], tp[ks * 12 + 1], tp[ks * 12 + 2], tap[ks]);
#pragma unroll
    for (int ks = 0; ks < CK; ks++)
#pragma unroll
        for (int t = 0; t < 5; t++) asm volatile("" : "+v"(tap[ks][t]));
}

// One input row of a stage in the operand layout.  src: LDS byte address of this lane's left neighbour pixel, channel kq, of the row
// image (pixel stride PSV floats; tile nt is 16 pixels on); its ky = 2 / 1 / 0 taps go to the partial depthwise rows r-1 / r / r+1
// (aPN on entry / aC / aPN on exit).  With EMIT the finished depthwise row r-1 is the B operand of this row's MFMAs into D
// (A operands: LDS byte address aop + lane, [ks][mt][64]).
// The LDS reads are inline asm with their own waits: left to the compiler, the reads of a whole row are merged across k-steps and
// hoisted (the taps get spilled).  Every wait is lgkmcnt(0): scalar loads share the counter and return out of order.
// TAPL: the taps are read from LDS per k-step (tapl: this lane's [ks][.][12] block) instead of living in registers (tap unused).
template <int CK, int MT, int WT, int PSV, bool EMIT, bool TAPL = false>
__device__ __forceinline__ void mdb_row(const unsigned src, const unsigned aop, const dv2f (&tap)[TAPL ? 1 : CK][5], RowAcc<WT> (&aPN)[CK], RowAcc<WT> (&aC)[CK],
                                        df32x4 (&D)[MT][WT], const float4* tapl = nullptr) {
    constexpr int NP = WT / 2, ODD = WT & 1;
    float xs[2][3][WT], av[2][MT];
    auto load_ks = [&](auto ksc, float (&x)[3][WT], float (&aw)[MT]) {
        constexpr int ks = decltype(ksc)::value;
        const unsigned xa = src, aa = aop;
#pragma unroll
        for (int dx = 0; dx < 3; dx++)
#pragma unroll
            for (int nt = 0; nt < WT; nt++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[dx][nt]) : "v"(xa), "n"((dx * PSV + 4 * ks) * 4 + nt * 16 * PSV * 4));
        if constexpr (EMIT) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(aw[mt]) : "v"(aa), "n"((ks * MT + mt) * 64 * 4));
        }
    };
    // The loaded registers pass through an (empty) asm statement behind the wait: their consumers then depend on something that is ordered
    // behind the s_waitcnt, whatever a later compiler would like to hoist (to the compiler the ds_read asm "returned" its value at once).
    auto landed = [&](float (&x)[3][WT], float (&aw)[MT]) {
#pragma unroll
        for (int dx = 0; dx < 3; dx++)
#pragma unroll
            for (int nt = 0; nt < WT; nt++) asm volatile("" : "+v"(x[dx][nt]));
        if constexpr (EMIT) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) asm volatile("" : "+v"(aw[mt]));
        }
    };
    auto kstep = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        if constexpr (ks + 1 < CK) load_ks(std::integral_constant<int, ks + 1>{}, xs[(ks + 1) & 1], av[(ks + 1) & 1]);
        float (&x)[3][WT] = xs[ks & 1];
        float (&aw)[MT] = av[ks & 1];
        dv2f w[5];
        if constexpr (TAPL) {
            dtaps_from(tapl[ks * 12], tapl[ks * 12 + 1], tapl[ks * 12 + 2], w);
        } else {
#pragma unroll
            for (int t = 0; t < 5; t++) w[t] = tap[ks][t];
        }
        float pch[WT];
        // tile pairs: tap t = half (t & 1) of w[t >> 1]; per accumulator the same order of operations as the single-tile form below
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const dv2f x0 = dv2f{x[0][2 * q], x[0][2 * q + 1]}, x1 = dv2f{x[1][2 * q], x[1][2 * q + 1]}, x2 = dv2f{x[2][2 * q], x[2][2 * q + 1]};
            dv2f c = aC[ks].p[q], pc = aPN[ks].p[q];
            if (EMIT) { dpk_fma<0>(pc, x0, w[3]); dpk_fma<1>(pc, x1, w[3]); dpk_fma<0>(pc, x2, w[4]); }
            dv2f n = dpk_mul<0>(x0, w[0]);
            dpk_fma<1>(n, x1, w[0]);
            dpk_fma<0>(n, x2, w[1]);
            dpk_fma<1>(c, x0, w[1]); dpk_fma<0>(c, x1, w[2]); dpk_fma<1>(c, x2, w[2]);
            aC[ks].p[q] = c;
            aPN[ks].p[q] = n;
            asm volatile("" : "+v"(aC[ks].p[q]), "+v"(aPN[ks].p[q]));  // pinned: LLVM would sink these updates into the next row
            pch[2 * q] = pc.x;
            pch[2 * q + 1] = pc.y;
        }
        if constexpr (ODD) {
            constexpr int nt = WT - 1;
            float n = x[0][nt] * w[0].x, c = aC[ks].s[0], pc = aPN[ks].s[0];
            if (EMIT) { pc = __builtin_fmaf(x[0][nt], w[3].x, pc); pc = __builtin_fmaf(x[1][nt], w[3].y, pc); pc = __builtin_fmaf(x[2][nt], w[4].x, pc); }
            n = __builtin_fmaf(x[1][nt], w[0].y, n);
            n = __builtin_fmaf(x[2][nt], w[1].x, n);
            c = __builtin_fmaf(x[0][nt], w[1].y, c);
            c = __builtin_fmaf(x[1][nt], w[2].x, c);
            c = __builtin_fmaf(x[2][nt], w[2].y, c);
            aC[ks].s[0] = c;
            aPN[ks].s[0] = n;
            asm volatile("" : "+v"(aC[ks].s[0]), "+v"(aPN[ks].s[0]));
            pch[nt] = pc;
        }
        if constexpr (EMIT) {
            // The packed FMAs are inline asm, invisible to the compiler's hazard recogniser, and v_mfma reads a register a v_pk_fma_f32 wrote
            // fewer than two instructions earlier STALE (tools/pk_mfma_hazard.hip: B = old value at 0 / 1 instructions in between, new from 2
            // on).  The asm statements are volatile (program order: a pair's finished row first, then its six other operations) and
            // nothing crosses this fence, so at least six instructions lie between the last write of a B operand and the first MFMA.
            if constexpr (NP > 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < WT; nt++) D[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt], pch[nt], D[mt][nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next k-step's operands (a whole k-step of cover)
        if constexpr (ks + 1 < CK) landed(xs[(ks + 1) & 1], av[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
    };
    load_ks(std::integral_constant<int, 0>{}, xs[0], av[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    landed(xs[0], av[0]);
    __builtin_amdgcn_sched_barrier(0);
    dfor_each(kstep, std::make_integer_sequence<int, CK>{});
}

}  // namespace

}  // namespace mi
