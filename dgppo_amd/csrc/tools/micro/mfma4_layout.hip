// Prints the operand layout of v_mfma_f32_4x4x1_16B_f32 on this device: which A lane and which B lane feed register r of the
// D value held by each lane.  (The block-diagonal attention kernels in csrc/nn_graph.hip assume: block = lane / 4, A row =
// lane % 4, B column = lane % 4, D[row r][col lane % 4] in register r of the lane.)
//   hipcc --offload-arch=gfx950 -O2 -o mfma4_layout tools/micro/mfma4_layout.hip && ./mfma4_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int lane = threadIdx.x;
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)lane, 1.0f, z, 0, 0, 0);
  f32x4 db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)lane, z, 0, 0, 0);
  for (int r = 0; r < 4; ++r) { out[lane * 8 + r] = da[r]; out[lane * 8 + 4 + r] = db[r]; }
}
int main() {
  float* d; float h[64 * 8];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d: A lanes [%2.0f %2.0f %2.0f %2.0f]  B lanes [%2.0f %2.0f %2.0f %2.0f]\n", l, h[l*8], h[l*8+1], h[l*8+2], h[l*8+3], h[l*8+4], h[l*8+5], h[l*8+6], h[l*8+7]);
    for (int r = 0; r < 4; ++r) { if ((int)h[l*8+r] != (l / 4) * 4 + r) ok = 0; if ((int)h[l*8+4+r] != l) ok = 0; }
  }
  printf("assumed layout %s\n", ok ? "CONFIRMED" : "WRONG");
  return ok ? 0 : 1;
}
