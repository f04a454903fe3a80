// Standalone probe: does the accumulation-chain shape limit the v_mfma_f32_32x32x16_bf16 rate?  Operands in registers,
// random data, 256 workgroups.  Variants: NCH independent accumulators visited round-robin (a dependent MFMA follows
// NCH-1 independent ones).  NCH=1: one serial chain; 2: the S^T/dP shape; 4: the PV / dK,dV shape; 8: no dependence in sight.
// hipcc --offload-arch=gfx950 -O3 mfma_chain_probe.hip -o chain.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2);} } while (0)

template <int NT, int NCH>
__global__ __launch_bounds__(NT) void probe(const float* in, float* out, int iters) {
  bf16x8 a[8], b[8];
  for (int s = 0; s < 8; ++s) for (int j = 0; j < 8; ++j) { a[s][j] = (__bf16)in[(threadIdx.x * 64 + s * 8 + j) & 4095]; b[s][j] = (__bf16)in[(threadIdx.x * 64 + s * 8 + j + 2048) & 4095]; }
  f32x16 O[NCH];
  for (int d = 0; d < NCH; ++d) for (int g = 0; g < 16; ++g) O[d][g] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 32 / NCH; ++s)
#pragma unroll
      for (int d = 0; d < NCH; ++d) O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + d) & 7], b[s & 7], O[d], 0, 0, 0);
  }
  float acc = 0.f; for (int d = 0; d < NCH; ++d) for (int g = 0; g < 16; ++g) acc += O[d][g];
  out[blockIdx.x * NT + threadIdx.x] = acc;
}

// same stream with v_mfma_f32_16x16x32_bf16 (half the FLOPs per instruction, 4 accumulator registers): 2*NCH accumulators
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NT, int NCH>
__global__ __launch_bounds__(NT) void probe16(const float* in, float* out, int iters) {
  bf16x8 a[8], b[8];
  for (int s = 0; s < 8; ++s) for (int j = 0; j < 8; ++j) { a[s][j] = (__bf16)in[(threadIdx.x * 64 + s * 8 + j) & 4095]; b[s][j] = (__bf16)in[(threadIdx.x * 64 + s * 8 + j + 2048) & 4095]; }
  f32x4 O[2 * NCH];
  for (int d = 0; d < 2 * NCH; ++d) for (int g = 0; g < 4; ++g) O[d][g] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 32 / NCH; ++s)
#pragma unroll
      for (int d = 0; d < 2 * NCH; ++d) O[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(s + d) & 7], b[s & 7], O[d], 0, 0, 0);
  }
  float acc = 0.f; for (int d = 0; d < 2 * NCH; ++d) for (int g = 0; g < 4; ++g) acc += O[d][g];
  out[blockIdx.x * NT + threadIdx.x] = acc;
}
template <int NT, int NCH> void run16(const float* din, float* dout, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 4000, blocks = 256;
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    probe16<NT, NCH><<<blocks, NT>>>(din, dout, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  const double flops = (double)blocks * (NT / 64) * iters * 64.0 * 2 * 16 * 16 * 32;
  printf("16x16x32  waves/SIMD %d  chains %d: %.3f ms  %.0f TFLOP/s\n", NT / 256, 2 * NCH, best, flops / best / 1e9);
}

template <int NT, int NCH> void run(const float* din, float* dout, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 4000, blocks = 256;
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    probe<NT, NCH><<<blocks, NT>>>(din, dout, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  const double flops = (double)blocks * (NT / 64) * iters * 32.0 * 2 * 32 * 32 * 16;
  printf("waves/SIMD %d  chains %d: %.3f ms  %.0f TFLOP/s\n", NT / 256, NCH, best, flops / best / 1e9);
}

int main() {
  std::vector<float> h(4096); for (auto& x : h) x = (rand() % 2001 - 1000) / 1000.f;
  float *din, *dout; CK(hipMalloc(&din, 4096 * 4)); CK(hipMalloc(&dout, 1024 * 512 * 4));
  CK(hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  run<256, 1>(din, dout, e0, e1); run<256, 2>(din, dout, e0, e1); run<256, 4>(din, dout, e0, e1); run<256, 8>(din, dout, e0, e1);
  run<512, 1>(din, dout, e0, e1); run<512, 2>(din, dout, e0, e1); run<512, 4>(din, dout, e0, e1); run<512, 8>(din, dout, e0, e1);
  run16<256, 2>(din, dout, e0, e1); run16<256, 8>(din, dout, e0, e1); run16<512, 2>(din, dout, e0, e1); run16<512, 8>(din, dout, e0, e1);
  run<1024, 4>(din, dout, e0, e1); run16<1024, 4>(din, dout, e0, e1);
  return 0;
}
