// Issue-rate microbenchmark for the FMA forms the strip kernels are built from (gfx950).  Each wave runs LOOPS x 48 independent
// instructions of one form on 24 accumulator pairs; the grid puts 1 or 2 waves on every SIMD.  Prints cycles per instruction per
// wave (s_memtime) and the chip-wide TFLOP/s.   Build: hipcc -O3 --offload-arch=gfx950 tools/valu_bench.hip -o tools/bin/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP12(x) REP4(x) REP4(x) REP4(x)

// FORM 0: v_pk_fma_f32 v, v, v, v        1: v_pk_fma_f32 v, v, s[pair], v      2: v_pk_fma_f32 v, v(op_sel broadcast), s[pair], v
//      3: v_fma_f32 v, v, v, v           4: v_fma_f32 v, s, v, v               5: v_pk_fma_f32 with two DIFFERENT sgpr pairs alternating
//      6: v_pk_mul_f32 v, v, v           7: v_mfma_f32_32x32x2_f32 (for reference)
template <int FORM>
__global__ __launch_bounds__(256) void valu_kernel(float* out, unsigned long long* cyc, int loops, float s0, float s1) {
    v2f acc[24];
#pragma unroll
    for (int i = 0; i < 24; i++) acc[i] = v2f{(float)threadIdx.x * 0.001f + i, 1.0f};
    v2f x = v2f{(float)threadIdx.x * 1e-3f, 0.5f};
    v2f w = v2f{s0, s1}, w2 = v2f{s1, s0};
    asm volatile("" : "+s"(w), "+s"(w2));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < loops; l++) {
#pragma unroll
        for (int i = 0; i < 24; i++) {
            if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(x));
            if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "s"(w));
            if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "s"(w));
            if (FORM == 3) asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(x.x), "v"(x.y));
            if (FORM == 4) asm volatile("v_fma_f32 %0, %3, %2, %0\n\tv_fma_f32 %1, %4, %2, %1" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(x.x), "s"(w.x), "s"(w.y));
            if (FORM == 5) { if (i & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "s"(w)); else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "s"(w2)); }
            if (FORM == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(x));
        }
#pragma unroll
        for (int i = 0; i < 24; i++) {
            if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(x));
            if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "s"(w));
            if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "s"(w));
            if (FORM == 3) asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(x.x), "v"(x.y));
            if (FORM == 4) asm volatile("v_fma_f32 %0, %3, %2, %0\n\tv_fma_f32 %1, %4, %2, %1" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(x.x), "s"(w.x), "s"(w.y));
            if (FORM == 5) { if (i & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "s"(w)); else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "s"(w2)); }
            if (FORM == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(x));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
#pragma unroll
    for (int i = 0; i < 24; i++) r += acc[i].x + acc[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// MFMA forms: 12 independent 16x16 accumulator tiles (48 VGPRs), A / B operands in VGPRs.
// MODE 0: v_mfma_f32_16x16x4_f32 back to back; 1: each MFMA followed by one v_pk_fma_f32; 2: by two; 3: by one ds_read_b32 (address in LDS)
typedef float f4v __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void mfma_kernel(float* out, unsigned long long* cyc, int loops) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f4v acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = f4v{0.f, 0.f, 0.f, 0.f};
    v2f pa[12];
#pragma unroll
    for (int i = 0; i < 12; i++) pa[i] = v2f{(float)i, 1.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    v2f x = v2f{a, b};
    float r = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < loops; l++) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 1 || MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pa[i]) : "v"(x));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pa[(i + 6) % 12]) : "v"(x));
            if (MODE == 3) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((threadIdx.x & 255) * 4)); r += 0.f * 0; asm volatile("" :: "v"(t)); }
        }
        if (MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 12; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w + pa[i].x + pa[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}


// v_mfma_f32_4x4x1_16b_f32 (16 blocks of (4x1)(1x4); the pointwise conv of strip_pipe2_kernel<.., MF>): 12 independent accumulators.
// MODE 0: back to back, A broadcast from one block (cbsz:4 abid:n); 1: no broadcast; 2: a stage of the row pipeline — 18 v_pk_fma_f32
// with SGPR-pair operands, then 24 MFMAs whose B operands they produced; 3: the same, interleaved 3 : 4; 4: 18 pk + 24 MFMA, B not dependent
template <int MODE>
__global__ __launch_bounds__(256) void mfma4_kernel(float* out, unsigned long long* cyc, int loops, float s0, float s1) {
    f4v acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = f4v{0.f, 0.f, 0.f, 0.f};
    v2f pa[6];
#pragma unroll
    for (int i = 0; i < 6; i++) pa[i] = v2f{(float)i, 1.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    v2f x = v2f{a, b};
    v2f w = v2f{s0, s1};
    asm volatile("" : "+s"(w));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < loops; l++) {
        if (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    if (MODE == 0) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:5" : "+v"(acc[i]) : "v"(a), "v"(b));
                    else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                }
        } else if (MODE == 2 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < 18; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i % 6]) : "v"(x), "s"(w));
            asm volatile("s_nop 1");
#pragma unroll
            for (int i = 0; i < 24; i++) {
                if (MODE == 2) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:5" : "+v"(acc[i % 12]) : "v"(a), "v"(i < 12 ? pa[0].x : pa[0].y));
                else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:5" : "+v"(acc[i % 12]) : "v"(a), "v"(b));
            }
        } else {
#pragma unroll
            for (int g = 0; g < 6; g++) {
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:5" : "+v"(acc[(4 * g + i) % 12]) : "v"(a), "v"(b));
#pragma unroll
                for (int i = 0; i < 3; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[(3 * g + i) % 6]) : "v"(x), "s"(w));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 12; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
    for (int i = 0; i < 6; i++) r += pa[i].x + pa[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run_mfma4(const char* name, int pk_per_loop) {
    const int loops = 2000;
    for (int wps = 1; wps <= 2; wps++) {
        const int grid = 256 * wps;
        float* out; unsigned long long* cyc;
        CK(hipMalloc(&out, (size_t)grid * 256 * 4)); CK(hipMalloc(&cyc, (size_t)grid * 4 * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        mfma4_kernel<MODE><<<grid, 256>>>(out, cyc, loops, 1.0001f, 0.9999f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        mfma4_kernel<MODE><<<grid, 256>>>(out, cyc, loops, 1.0001f, 0.9999f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h((size_t)grid * 4);
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0; for (auto v : h) sum += (double)v;
        const double mf = (double)loops * 24;
        const double flop = (double)grid * 4 * loops * (24.0 * 512 + pk_per_loop * 256.0);
        printf("%-52s %d wave(s)/SIMD: %6.2f memtime ticks per loop (24 MFMA + %d pk) per wave, %7.1f TFLOP/s, %.3f ms, %.2f GHz by ticks\n", name, wps,
               sum / h.size() / loops, pk_per_loop, flop / ms / 1e9, ms, sum / h.size() / ms / 1e6);
        CK(hipFree(out)); CK(hipFree(cyc));
    }
}

template <int MODE>
void run_mfma(const char* name) {
    const int loops = 2000;
    for (int wps = 1; wps <= 2; wps++) {
        const int grid = 256 * wps;
        float* out; unsigned long long* cyc;
        CK(hipMalloc(&out, (size_t)grid * 256 * 4)); CK(hipMalloc(&cyc, (size_t)grid * 4 * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        mfma_kernel<MODE><<<grid, 256>>>(out, cyc, loops);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        mfma_kernel<MODE><<<grid, 256>>>(out, cyc, loops);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h((size_t)grid * 4);
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0; for (auto v : h) sum += (double)v;
        const double mf = (double)loops * 12;
        printf("%-52s %d wave(s)/SIMD: %6.2f memtime ticks per MFMA per wave, %7.1f TFLOP/s (MFMA only), %.3f ms\n", name, wps, sum / h.size() / mf,
               (double)grid * 4 * mf * 16 * 16 * 4 * 2 / ms / 1e9, ms);
        CK(hipFree(out)); CK(hipFree(cyc));
    }
}

template <int FORM>
void run(const char* name, int flops_per_inst_lane, int insts_per_slot) {
    const int loops = 2000;
    for (int wps = 1; wps <= 2; wps++) {
        const int grid = 256 * wps;  // 256-thread workgroups: 4 waves = one per SIMD; wps workgroups per CU
        float* out; unsigned long long* cyc;
        CK(hipMalloc(&out, (size_t)grid * 256 * 4)); CK(hipMalloc(&cyc, (size_t)grid * 4 * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        valu_kernel<FORM><<<grid, 256>>>(out, cyc, loops, 1.0001f, 0.9999f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        valu_kernel<FORM><<<grid, 256>>>(out, cyc, loops, 1.0001f, 0.9999f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h((size_t)grid * 4);
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0; for (auto v : h) sum += (double)v;
        const double insts = (double)loops * 48 * insts_per_slot;
        const double flop = (double)grid * 4 * insts * 64 * flops_per_inst_lane;
        printf("%-52s %d wave(s)/SIMD: %6.2f memtime ticks per instruction per wave, %7.1f TFLOP/s, %.3f ms\n", name, wps, sum / h.size() / insts, flop / ms / 1e9, ms);
        CK(hipFree(out)); CK(hipFree(cyc));
    }
}

int main() {
    run<0>("v_pk_fma_f32 v, v, v, v", 4, 1);
    run<1>("v_pk_fma_f32 v, v, s[2], v", 4, 1);
    run<2>("v_pk_fma_f32 v, v.x bcast, s[2], v", 4, 1);
    run<5>("v_pk_fma_f32 v, v.x bcast, alternating s[2]", 4, 1);
    run<3>("v_fma_f32 v, v, v, v (x2)", 2, 2);
    run<4>("v_fma_f32 v, s, v, v (x2)", 2, 2);
    run<6>("v_pk_mul_f32 v, v, v", 2, 1);
    run_mfma<0>("v_mfma_f32_16x16x4_f32 x12 tiles");
    run_mfma<1>("v_mfma_f32_16x16x4_f32 + 1 v_pk_fma_f32 each");
    run_mfma<2>("v_mfma_f32_16x16x4_f32 + 2 v_pk_fma_f32 each");
    run_mfma<3>("v_mfma_f32_16x16x4_f32 + 1 ds_read_b32 each");
    run_mfma4<0>("v_mfma_f32_4x4x1_16b_f32 cbsz:4 abid:5 x12 acc", 0);
    run_mfma4<1>("v_mfma_f32_4x4x1_16b_f32 (no broadcast) x12 acc", 0);
    run_mfma4<4>("18 v_pk_fma_f32 s[2], then 24 4x4x1 (independent B)", 18);
    run_mfma4<2>("18 v_pk_fma_f32 s[2], then 24 4x4x1 (B = their result)", 18);
    run_mfma4<3>("6 x (4 4x4x1, 3 v_pk_fma_f32 s[2])", 18);
    return 0;
}
