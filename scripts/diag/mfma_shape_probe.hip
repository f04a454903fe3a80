// Diagnostic (not part of the product library): sustained rate of the two bf16 MFMA shapes on RANDOM operands, operands in registers,
// one or two waves per SIMD.  Under load the chip lowers its clock (MI355X_MICROARCH.md "DVFS give-back"), so cycles per FLOP do not
// decide which shape is faster by wall time.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o build/libmfma_probe.so <this file>
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(256) void probe_kernel(const bf16x8* __restrict__ src, float* __restrict__ dst, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 0xffff]; b[i] = src[(tid * 8 + 4 + i) & 0xffff]; }
  if (SHAPE == 0) {                       // 32x32x16: 8 independent accumulators of 16 registers
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i + (i >> 2)) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) s += acc[i][g];
    dst[tid] = s;
  } else {                                // 16x16x32: 16 independent accumulators of 4 registers
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[i][g] = 0.f;
    for (int it = 0; it < iters; ++it) {                // 16 MFMAs of 16384 FLOP = the 8 x 32768 of the other shape
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i + (i >> 2)) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) s += acc[i][g];
    dst[tid] = s;
  }
}

// returns milliseconds per launch (average over reps) or a negative HIP error code; flops per launch = blocks*4 waves * iters * 8 * 32768 (both shapes)
extern "C" float mfma_probe(int shape, int blocks, int iters, int reps, const void* src, void* dst) {
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
  for (int r = 0; r < reps + 3; ++r) {
    if (r == 3) hipEventRecord(e0, 0);
    if (shape == 0) hipLaunchKernelGGL(probe_kernel<0>, dim3(blocks), dim3(256), 0, 0, (const bf16x8*)src, (float*)dst, iters);
    else hipLaunchKernelGGL(probe_kernel<1>, dim3(blocks), dim3(256), 0, 0, (const bf16x8*)src, (float*)dst, iters);
  }
  hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return -2.f;
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps;
}
