// Development probe: do two small grids on two streams overlap on this box?  A kernel of G workgroups spins for ~T us.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__global__ void spin(long long ticks, int* sink) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (ticks < 0) *sink = 1;
}
int main(int argc, char** argv) {
    int G = argc > 1 ? atoi(argv[1]) : 64, threads = argc > 2 ? atoi(argv[2]) : 512, lds = argc > 3 ? atoi(argv[3]) : 0;
    int* sink; CK(hipMalloc(&sink, 4));
    hipStream_t s[4];
    for (int i = 0; i < 4; i++) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    if (lds > 64 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const long long ticks = 100 * 100;  // wall_clock64: 100 MHz -> 100 us
    for (int L = 1; L <= 4; L *= 2) {
        for (int w = 0; w < 2; w++) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int rep = 0; rep < 20; rep++)
                for (int l = 0; l < L; l++) spin<<<G, threads, lds, s[l]>>>(ticks, sink);
            CK(hipDeviceSynchronize());
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (w) printf("grid %d x %d threads, %d B LDS, %d stream(s): %.3f ms for 20 x %d launches of 0.1 ms (serial = %.1f ms)\n", G, threads, lds, L, ms, L, 2.0 * L);
        }
    }
    return 0;
}
