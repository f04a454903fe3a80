// Standalone probe: sustained v_mfma_f32_32x32x16_bf16 rate with the attention kernels' chain shape
// (two 8-deep accumulation chains, then 16 MFMAs over four accumulators), operands in registers, random data,
// 8 waves per workgroup (2 per SIMD) or 4 (1 per SIMD), one workgroup per CU slot.  Gives the ceiling the
// tile loops are measured against.   hipcc --offload-arch=gfx950 -O3 mfma_rate_probe.hip -o rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2);} } while (0)

template <int NT>
__global__ __launch_bounds__(NT) void probe(const float* in, float* out, int iters) {
  bf16x8 a[8], b[8];
  for (int s = 0; s < 8; ++s) for (int j = 0; j < 8; ++j) { a[s][j] = (__bf16)in[(threadIdx.x * 64 + s * 8 + j) & 4095]; b[s][j] = (__bf16)in[(threadIdx.x * 64 + s * 8 + j + 2048) & 4095]; }
  f32x16 O[4];
  for (int d = 0; d < 4; ++d) for (int g = 0; g < 16; ++g) O[d][g] = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x16 X[2];
    for (int kb = 0; kb < 2; ++kb) { for (int g = 0; g < 16; ++g) X[kb][g] = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) X[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[(s + kb) & 7], X[kb], 0, 0, 0); }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      bf16x8 pb;
      for (int j = 0; j < 8; ++j) pb[j] = (__bf16)(X[s4 >> 1][8 * (s4 & 1) + j] * 1e-3f);
#pragma unroll
      for (int d = 0; d < 4; ++d) O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(d + s4) & 7], pb, O[d], 0, 0, 0);
    }
  }
  float acc = 0.f; for (int d = 0; d < 4; ++d) for (int g = 0; g < 16; ++g) acc += O[d][g];
  out[blockIdx.x * NT + threadIdx.x] = acc;
}

int main() {
  std::vector<float> h(4096); for (auto& x : h) x = (rand() % 2001 - 1000) / 1000.f;
  float *din, *dout; CK(hipMalloc(&din, 4096 * 4)); CK(hipMalloc(&dout, 1024 * 512 * 4));
  CK(hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  for (int nt : {512, 256}) {
    for (int rep = 0; rep < 3; ++rep) {
      const int blocks = 256 * (nt == 512 ? 1 : 1);
      CK(hipEventRecord(e0));
      if (nt == 512) probe<512><<<blocks, 512>>>(din, dout, iters); else probe<256><<<blocks, 256>>>(din, dout, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double flops = (double)blocks * (nt / 64) * iters * 32.0 * 2 * 32 * 32 * 16;
      printf("threads/WG %d (%d waves/SIMD): %.3f ms  %.0f TFLOP/s\n", nt, nt / 256, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
