// TEST INFRASTRUCTURE ONLY: storage of the emulated launch indices (see hip/hip_runtime.h next to this file).
#include <hip/hip_runtime.h>
thread_local dgppo_emul_idx blockIdx, threadIdx, blockDim, gridDim;
