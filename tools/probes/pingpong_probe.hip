// pingpong_probe.hip — latency of a tagged-packet hand-over between TWO workgroups of one launch (the halo exchange of bandnet_kernels.hip), by where
// the two sit and how the packet travels.  Workgroup A = blockIdx 0; B = blockIdx `partner` (8: the same XCD as A under round-robin dispatch, 1: the next
// XCD).  A stores {value, tag r}, B polls until it sees tag r and answers with tag r; R rounds; microseconds per ONE-WAY hand-over = time / (2 R).
//   store: 0 plain (write-back into the XCD's L2)  1 sc0 sc1 (write-through)      load: 0 sc0 (served by this XCD's L2)  1 sc0 sc1 (served by memory)
// A reader that never sees its tag gives up after a bounded number of polls (fail = 1): expected for plain stores / sc0 loads ACROSS XCDs (non-coherent L2s).
// hipcc -O3 --offload-arch=gfx950 tools/probes/pingpong_probe.hip -o tools/probes/_bin/pingpong_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int ST, int LD>
__global__ __launch_bounds__(64) void pp(unsigned* buf, int partner, int rounds, unsigned epoch, int* fail, unsigned* xcc) {
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == (unsigned)partner ? 1 : -1);
    if (threadIdx.x == 0 && me >= 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[me] = id;
    }
    if (me < 0) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1 << 20, 0x00020000);
    // lane l owns packet l (16 bytes each): 64 packets = 1 KB per direction, like one halo row of a narrow layer
    const int mine = (me * 64 + threadIdx.x) * 16, theirs = ((1 - me) * 64 + threadIdx.x) * 16;
    for (int r = 0; r < rounds; r++) {
        const unsigned tag = epoch + r + 1;
        if (me == 0 || r >= 0) {
            if (me == 1) {   // B waits for A's packet of this round first
                int it = 0;
                for (;;) {
                    u4 v = LD ? __builtin_amdgcn_raw_buffer_load_b128(rs, theirs, 0, 17) : __builtin_amdgcn_raw_buffer_load_b128(rs, theirs, 0, 1);
                    if (v.y == tag && v.w == tag) break;
                    if (++it > 200000) { *fail = 1; return; }
                }
            }
            u4 q = {(unsigned)r, tag, (unsigned)threadIdx.x, tag};
            if (ST) __builtin_amdgcn_raw_buffer_store_b128(q, rs, mine, 0, 17); else __builtin_amdgcn_raw_buffer_store_b128(q, rs, mine, 0, 0);
            if (me == 0) {   // A waits for B's answer
                int it = 0;
                for (;;) {
                    u4 v = LD ? __builtin_amdgcn_raw_buffer_load_b128(rs, theirs, 0, 17) : __builtin_amdgcn_raw_buffer_load_b128(rs, theirs, 0, 1);
                    if (v.y == tag && v.w == tag) break;
                    if (++it > 200000) { *fail = 1; return; }
                }
            }
        }
    }
}

int main() {
    unsigned *buf, *xcc; int* fail;
    CK(hipMalloc(&buf, 1 << 20)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&xcc, 8));
    CK(hipMemset(buf, 0, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rounds = 2000;
    unsigned epoch = 0;
    for (int partner : {8, 1, 16, 4})
        for (int st = 0; st < 2; st++)
            for (int ld = 0; ld < 2; ld++) {
                CK(hipMemset(fail, 0, 4));
                float best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0));
                    auto k = st ? (ld ? pp<1, 1> : pp<1, 0>) : (ld ? pp<0, 1> : pp<0, 0>);
                    hipLaunchKernelGGL(k, dim3(32), dim3(64), 0, 0, buf, partner, rounds, epoch, fail, xcc);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    epoch += rounds + 1;
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                int f; unsigned x[2];
                CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
                printf("partner block %2d (XCC %u -> %u)  store %-7s load %-7s : %7.3f us per one-way hand-over%s\n", partner, x[0] & 15, x[1] & 15, st ? "sc0 sc1" : "plain", ld ? "sc0 sc1" : "sc0",
                       best * 1000.f / (2 * rounds), f ? "   (GAVE UP: packets never seen)" : "");
            }
    return 0;
}
