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

// One input row of a stage in the operand layout.  src: LDS byte address of this lane's left neighbour pixel, channel kq, of the row
// image (pixel stride PSV floats; tile nt is 16 pixels on); its ky = 2 / 1 / 0 taps go to the partial depthwise rows r-1 / r / r+1
// (aPN on entry / aC / aPN on exit).  With EMIT the finished depthwise row r-1 is the B operand of this row's MFMAs into D
// (A operands: LDS byte address aop + lane, [ks][mt][64]).
// The LDS reads are inline asm with their own waits: left to the compiler, the reads of a whole row are merged across k-steps and
// hoisted (the taps get spilled).  Every wait is lgkmcnt(0): scalar loads share the counter and return out of order.
// TAPL: the taps are read from LDS per k-step (tapl: this lane's [ks][.][12] block) instead of living in registers (tap unused).
template <int CK, int MT, int WT, int PSV, bool EMIT, bool TAPL = false>
__device__ __forceinline__ void mdb_row(const unsigned src, const unsigned aop, const float (&tap)[TAPL ? 1 : CK][9], float (&aPN)[CK][WT], float (&aC)[CK][WT],
                                        df32x4 (&D)[MT][WT], const float4* tapl = nullptr) {
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
    auto kstep = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        if constexpr (ks + 1 < CK) load_ks(std::integral_constant<int, ks + 1>{}, xs[(ks + 1) & 1], av[(ks + 1) & 1]);
        float (&x)[3][WT] = xs[ks & 1];
        float (&aw)[MT] = av[ks & 1];
        float w[9];
        if constexpr (TAPL) {
            const float4 t0 = tapl[ks * 12], t1 = tapl[ks * 12 + 1], t2 = tapl[ks * 12 + 2];
            w[0] = t0.x; w[1] = t0.y; w[2] = t0.z; w[3] = t0.w; w[4] = t1.x; w[5] = t1.y; w[6] = t1.z; w[7] = t1.w; w[8] = t2.x;
        } else {
#pragma unroll
            for (int t = 0; t < 9; t++) w[t] = tap[ks][t];
        }
        float pch[WT];
#pragma unroll
        for (int nt = 0; nt < WT; nt++) {
            float n = x[0][nt] * w[0], c = aC[ks][nt], pc = aPN[ks][nt];
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                if (EMIT) pc = __builtin_fmaf(x[dx][nt], w[6 + dx], pc);
                if (dx) n = __builtin_fmaf(x[dx][nt], w[dx], n);
                c = __builtin_fmaf(x[dx][nt], w[3 + dx], c);
            }
            aC[ks][nt] = c;
            aPN[ks][nt] = n;
            asm volatile("" : "+v"(aC[ks][nt]), "+v"(aPN[ks][nt]));  // pinned: LLVM would sink these updates into the next row
            pch[nt] = pc;
        }
        if constexpr (EMIT) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < WT; nt++) D[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt], pch[nt], D[mt][nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next k-step's operands (a whole k-step of cover)
        __builtin_amdgcn_sched_barrier(0);
    };
    load_ks(std::integral_constant<int, 0>{}, xs[0], av[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    dfor_each(kstep, std::make_integer_sequence<int, CK>{});
}

}  // namespace

}  // namespace mi
