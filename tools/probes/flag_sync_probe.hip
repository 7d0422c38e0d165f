// flag_sync_probe.hip — how long does a hand-over between NEIGHBOURING workgroups of one launch take on gfx950?
// NW workgroups run R rounds of: write my band (bytes) to global memory, publish "round r done", wait for both neighbours' flags,
// read their bands.  Prints microseconds per round for several NW / band sizes and two publication protocols:
//   fence : plain stores, __threadfence(), agent-scope release store of the flag; acquire load of the neighbours' flags
//   sc1   : the band is stored and loaded with sc0 sc1 (write-through / bypass: coherent at the memory side), the flag likewise
// Every wait has an iteration bound (the launch always drains).   hipcc --offload-arch=gfx950 -O3 -o flag_sync_probe flag_sync_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1(float4* p, float4 v) {
    f4v x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ float4 ld_sc0(const float4* p) {
    f4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 ld_sc1(const float4* p) {
    f4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(float4* buf, int* flags, int f4_per_band, int rounds, int epoch, int* fail) {
    const int nw = gridDim.x, tid = threadIdx.x;
    // modes 2, 3: neighbouring bands on the same XCD (workgroup i is dispatched to XCD i % 8)
    const int w = (MODE >= 2 && nw % 8 == 0) ? (blockIdx.x % 8) * (nw / 8) + blockIdx.x / 8 : blockIdx.x;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int r = 0; r < rounds; r++) {
        float4* mine = buf + ((long)(r & 1) * nw + w) * f4_per_band;
        for (int i = tid; i < f4_per_band; i += 512) {
            float4 v = make_float4(r + acc.x, w, i, 1.f);
            if (MODE == 0) mine[i] = v; else st_sc1(mine + i, v);
        }
        if (MODE == 0) __threadfence(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + w, epoch + r + 1, MODE == 0 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 2) {
            const int nb = tid == 0 ? w - 1 : w + 1;
            if (nb >= 0 && nb < nw) {
                int it = 0;
                while (__hip_atomic_load(flags + nb, MODE == 0 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch + r + 1) {
                    if (++it > 2000000) { *fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        __syncthreads();
        if (MODE == 0) __threadfence();
        for (int d = -1; d <= 1; d += 2) {
            const int nb = w + d;
            if (nb < 0 || nb >= nw) continue;
            const float4* theirs = buf + ((long)(r & 1) * nw + nb) * f4_per_band;
            for (int i = tid; i < f4_per_band; i += 512) {
                float4 v = MODE == 0 ? theirs[i] : (MODE == 3 ? ld_sc0(theirs + i) : ld_sc1(theirs + i));
                acc.x += v.w * 1e-9f;
                if (v.y != (float)nb || v.z != (float)i) *fail = 2;
            }
        }
    }
    if (acc.x == 12345.f) buf[0] = acc;
}

int main() {
    const int rounds = 64;
    float4* buf; int *flags, *fail;
    CK(hipMalloc(&buf, 2L * 256 * 65536));
    CK(hipMalloc(&flags, 256 * sizeof(int)));
    CK(hipMalloc(&fail, sizeof(int)));
    CK(hipMemset(flags, 0, 256 * sizeof(int)));
    CK(hipMemset(fail, 0, sizeof(int)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int epoch = 0;
    for (int mode = 1; mode < 4; mode++)
        for (int nw : {8, 64, 128})
            for (int kb : {4, 24, 48}) {
                const int f4 = kb * 1024 / 16;
                float best = 1e9;
                for (int rep = 0; rep < 5; rep++) {
                    CK(hipEventRecord(e0));
                    if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(nw), dim3(512), 0, 0, buf, flags, f4, rounds, epoch, fail);
                    else if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(nw), dim3(512), 0, 0, buf, flags, f4, rounds, epoch, fail);
                    else hipLaunchKernelGGL(probe<3>, dim3(nw), dim3(512), 0, 0, buf, flags, f4, rounds, epoch, fail);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    epoch += rounds;
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                int f; CK(hipMemcpy(&f, fail, sizeof f, hipMemcpyDeviceToHost));
                printf("%s nw=%3d band=%2d KB: %.2f us/round (launch of %d rounds %.1f us) fail=%d\n", mode == 1 ? "sc1       " : (mode == 2 ? "sc1 +xcd  " : "sc0 +xcd  "), nw, kb, best * 1000.f / rounds, rounds, best * 1000.f, f);
            }
    return 0;
}
