// Round 6: issue rate of v_mfma_f32_4x4x1_16b_f32 streams like stem_mfma_kernel's (NACC accumulators round robin, A broadcast from one block of a weight
// register with cbsz = 4 / abid, the B register changing every NACC instructions).  Prints SIMD-level cycles per MFMA (s_memtime) for 1 / 2 / 4 waves per SIMD.
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma4_probe.hip -o /tmp/mfma4_probe && /tmp/mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NACC, int BCAST, int BVAR>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int loops) {
    v4f D[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) D[i] = v4f{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b[16];
#pragma unroll
    for (int i = 0; i < 16; i++) b[i] = threadIdx.x * 1e-4f + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < loops; l++) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
#pragma unroll
            for (int c = 0; c < NACC; c++) {
                const float bb = BVAR ? b[e] : b[0];
                if (BCAST) {
                    switch ((e * NACC + c) & 15) {
#define M(N) case N: D[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bb, D[c], 4, N, 0); break;
                        M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#undef M
                    }
                } else {
                    D[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bb, D[c], 0, 0, 0);
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r += D[i].x + D[i].y + D[i].z + D[i].w;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, int BCAST, int BVAR>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    const int loops = 400;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;   // 256 CUs x wps blocks of four waves: wps waves per SIMD
        CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&cyc, (size_t)blocks * 4 * 8));
        for (int rep = 0; rep < 3; rep++) k<NACC, BCAST, BVAR><<<blocks, 256>>>(out, cyc, loops);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(blocks * 4);
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double per_wave = (double)h[h.size() / 2] / (loops * 16.0 * NACC);
        printf("%-44s waves/SIMD %d: %6.2f cycles per MFMA per wave -> %6.2f per MFMA on the SIMD\n", name, wps, per_wave, per_wave / wps);
        CK(hipFree(out)); CK(hipFree(cyc));
    }
}

int main() {
    run<6, 1, 1>("6 accumulators, broadcast A, B varies");
    run<6, 1, 0>("6 accumulators, broadcast A, B fixed");
    run<6, 0, 1>("6 accumulators, no broadcast, B varies");
    run<12, 1, 1>("12 accumulators, broadcast A, B varies");
    run<24, 1, 1>("24 accumulators, broadcast A, B varies");
    run<3, 1, 1>("3 accumulators, broadcast A, B varies");
    return 0;
}
