// Development probe: does v_mfma_f32_16x16x4_f32 see the result of a v_pk_fma_f32 issued K instructions earlier (B operand = the low /
// high half of the packed result)?  One asm block per K so the compiler cannot reschedule; prints which value each MFMA used.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int K>
__global__ void probe(float* out) {
    // v[20:21] = packed accumulator (old value 1.0, 1.0); v[22:23] = x (2, 3); v[24:25] = w (5, 7) -> new = 1 + 2*5 = 11 (lo), 1 + 3*5 = 16 (hi)
    // A operand v26 = 1.0 in every lane -> D[i][j] = sum_k A[i][k] * B[k][j] = 4 lanes' worth of B summed over k (all lanes hold the same B) = 4 * B
    float dlo, dhi;
    asm volatile(
        "v_mov_b32 v20, 1.0\n\tv_mov_b32 v21, 1.0\n\tv_mov_b32 v22, 2.0\n\tv_mov_b32 v23, 0x40400000\n\tv_mov_b32 v24, 0x40a00000\n\tv_mov_b32 v25, 0x40e00000\n\t"
        "v_mov_b32 v26, 1.0\n\t"
        "v_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t"
        "v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\t"
        "s_nop 7\n\ts_nop 7\n\t"
        "v_pk_fma_f32 v[20:21], v[22:23], v[24:25], v[20:21] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        ".rept %2\n\tv_mov_b32 v36, v36\n\t.endr\n\t"
        "v_mfma_f32_16x16x4_f32 v[28:31], v26, v20, v[28:31]\n\t"
        "v_mfma_f32_16x16x4_f32 v[32:35], v26, v21, v[32:35]\n\t"
        "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
        "v_mov_b32 %0, v28\n\tv_mov_b32 %1, v32\n\t"
        : "=v"(dlo), "=v"(dhi) : "n"(K)
        : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36");
    if (threadIdx.x == 0) { out[2 * K] = dlo; out[2 * K + 1] = dhi; }
}

int main() {
    float* d; CK(hipMalloc(&d, 64 * 4)); CK(hipMemset(d, 0, 64 * 4));
    probe<0><<<1, 64>>>(d); probe<1><<<1, 64>>>(d); probe<2><<<1, 64>>>(d); probe<3><<<1, 64>>>(d); probe<4><<<1, 64>>>(d);
    probe<5><<<1, 64>>>(d); probe<6><<<1, 64>>>(d); probe<8><<<1, 64>>>(d);
    CK(hipDeviceSynchronize());
    float h[64]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    for (int k : {0, 1, 2, 3, 4, 5, 6, 8}) printf("K = %d independent VALU ops between: low-half MFMA saw B = %g (new 11, old 1), high-half MFMA saw B = %g (new 16, old 1)\n", k, h[2 * k] / 4, h[2 * k + 1] / 4);
    return 0;
}
