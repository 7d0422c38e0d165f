// block_kernels.hip — fused BlazeBlock kernels (placeholder until the MFMA kernel lands in this file).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace mi {
void block_weight_dims(int C, int Co, int* Cp, int* Cop) {
    *Cp = (C + 3) & ~3;
    *Cop = (Co + 3) & ~3;
}
bool block_kernel_supports(const BlockArgs&) { return false; }
int launch_block(const BlockArgs&, void*) { return (int)hipErrorNotSupported; }
}  // namespace mi
