// TEST INFRASTRUCTURE ONLY — never on the product path (libdgppo_hip.so is built by hipcc against the real HIP runtime).
//
// A serial host stand-in for the handful of HIP constructs that the THREAD-INDEPENDENT kernels of this library use
// (csrc/env_reset.hip: one thread per environment; the noise generators), so that their bodies can be compiled by g++ with
// -fsanitize=address,undefined and run in the GPU-less container (SURVEY §7 "one source, two targets", §5 "sanitizers";
// GPU AddressSanitizer is not available on the pool).  The round-1 abort was exactly in this code: private arrays
// indexed by loop variables.  Found on the include path BEFORE the ROCm headers by `make -C dgppo_amd/csrc cpu_asan`.
//
// Semantics: hipLaunchKernelGGL runs the grid serially, block by block, thread by thread, each thread to completion.  That is
// equivalent to the device for kernels without barriers, cross-lane operations or inter-thread communication; `__shared__`
// becomes one static array that all emulated threads see (env_reset_variant_kernel only uses its own row of it); atomicAdd
// is a plain add.  Kernels that synchronise threads must NOT be built this way.
#pragma once
#ifndef _GNU_SOURCE
#define _GNU_SOURCE            // sincosf
#endif
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct dgppo_emul_idx { unsigned x, y, z; };
extern thread_local dgppo_emul_idx blockIdx, threadIdx, blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "host emulation: no error"; }

template <typename T> static inline T atomicAdd(T* p, T v) { T old = *p; *p = old + v; return old; }

#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...)                        \
  do {                                                                                    \
    const dim3 g__ = (grid), b__ = (block);                                               \
    (void)(smem); (void)(stream);                                                         \
    gridDim = {g__.x, g__.y, g__.z}; blockDim = {b__.x, b__.y, b__.z};                    \
    for (unsigned bx__ = 0; bx__ < g__.x; ++bx__)                                         \
      for (unsigned tx__ = 0; tx__ < b__.x; ++tx__) {                                     \
        blockIdx = {bx__, 0, 0}; threadIdx = {tx__, 0, 0};                                \
        kernel(__VA_ARGS__);                                                              \
      }                                                                                   \
  } while (0)
