// What clock does the chip hold under each instruction stream?  s_memtime (shader-clock ticks) against s_memrealtime (constant 100 MHz) inside
// one long kernel on every SIMD (1 or 2 waves per SIMD): MHz = 100 x d(memtime) / d(memrealtime); chip-wide TFLOP/s from HIP events.
// Round 6: the row pipelines' step time did not move when barriers / hand-over writes were removed (profiles/r06_pipe_ablations.txt) —
// is the chip clock- (power-) limited under the MFMA + packed-FMA mix?   Build: hipcc -O3 --offload-arch=gfx950 tools/probes/clock_probe.hip -o tools/probes/_bin/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// MODE 0: s_sleep only   1: v_mfma_f32_4x4x1_16b_f32 stream   2: v_mfma_f32_16x16x4_f32 stream   3: v_pk_fma_f32 stream
//      4: the pipelines' mix (18 packed FMAs, then 24 4x4x1 MFMAs)   5: mix with 16x16x4 (18 pk, then 6)   6: v_mfma_f32_32x32x2_f32
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* st, int loops) {
    v4f acc[12];
    v2f pa[12];
    typedef float v16f __attribute__((ext_vector_type(16)));
    v16f big[2];
#pragma unroll
    for (int i = 0; i < 12; i++) { acc[i] = v4f{0.f, 0.f, 0.f, 0.f}; pa[i] = v2f{(float)i, 1.f}; }
#pragma unroll
    for (int i = 0; i < 16; i++) { big[0][i] = 0.f; big[1][i] = 0.f; }
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    v2f x = v2f{a, b};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int l = 0; l < loops; l++) {
        if (MODE == 0) { __builtin_amdgcn_s_sleep(127); }
        if (MODE == 1) {
#pragma unroll
            for (int rep = 0; rep < 4; rep++)
#pragma unroll
                for (int i = 0; i < 12; i++) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:3" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 12; i++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        if (MODE == 3) {
#pragma unroll
            for (int rep = 0; rep < 4; rep++)
#pragma unroll
                for (int i = 0; i < 12; i++) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pa[i]) : "v"(x));
        }
        if (MODE == 4 || MODE == 5) {
#pragma unroll
            for (int i = 0; i < 18; i++) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pa[i % 12]) : "v"(x));
            asm volatile("s_nop 1");
            if (MODE == 4) {
#pragma unroll
                for (int rep = 0; rep < 2; rep++)
#pragma unroll
                    for (int i = 0; i < 12; i++) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:3" : "+v"(acc[i]) : "v"(pa[i].x), "v"(b));
            } else {
#pragma unroll
                for (int i = 0; i < 6; i++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(pa[i].x), "v"(b));
            }
        }
        if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 2; i++) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(big[i]) : "v"(a), "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w + pa[i].x + pa[i].y;
    r += big[0][0] + big[1][5];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) { st[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; st[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, double flops_per_loop_per_wave, int wps, int loops) {
    const int wgs = 256 * wps;  // 256 CUs x (4 waves per workgroup = 1 per SIMD) x wps
    float* out; unsigned long long* st;
    CK(hipMalloc(&out, (size_t)wgs * 256 * 4)); CK(hipMalloc(&st, (size_t)wgs * 4 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<wgs, 256>>>(out, st, loops / 8);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<MODE><<<wgs, 256>>>(out, st, loops);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)wgs * 8);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    double t = 0, r = 0;
    for (size_t i = 0; i < h.size(); i += 2) { t += h[i]; r += h[i + 1]; }
    const double waves = wgs * 4.0;
    printf("%-44s %d wave/SIMD: %8.3f ms  memtime/realtime %7.3f -> %6.0f MHz if memtime = shader clock;  %7.1f TFLOP/s  (%.1f memtime ticks per loop per wave)\n", name, wps, ms, t / r, 100.0 * t / r,
           flops_per_loop_per_wave * loops * waves / ms / 1e9, t / waves / loops);
    CK(hipFree(out)); CK(hipFree(st));
}

int main(int argc, char** argv) {
    const int loops = argc > 1 ? atoi(argv[1]) : 20000;
    for (int wps = 1; wps <= 2; wps++) {
        run<0>("s_sleep 127", 0, wps, loops / 20);
        run<1>("v_mfma_f32_4x4x1_16b_f32 x48", 48 * 512.0, wps, loops);
        run<2>("v_mfma_f32_16x16x4_f32 x12", 12 * 2048.0, wps, loops);
        run<6>("v_mfma_f32_32x32x2_f32 x2", 2 * 4096.0, wps, loops * 3);
        run<3>("v_pk_fma_f32 x48", 48 * 256.0, wps, loops);
        run<4>("18 pk_fma + 24 mfma 4x4x1", 18 * 256.0 + 24 * 512.0, wps, loops * 2);
        run<5>("18 pk_fma + 6 mfma 16x16x4", 18 * 256.0 + 6 * 2048.0, wps, loops * 2);
    }
    return 0;
}
