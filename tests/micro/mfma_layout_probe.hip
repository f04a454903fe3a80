// Standalone probe (not part of the product): checks on real gfx950 hardware the lane maps the
// attention kernels rely on.  Build: hipcc --offload-arch=gfx950 -O2 mfma_layout_probe.hip -o probe
//   1. v_mfma_f32_32x32x16_bf16 A/B/C maps (asymmetric integer data)
//   2. accumulator tile X re-used as the B operand (A*X) with the permuted k order
//   3. ds_read_b64_tr_b16 semantics as used for the transposed (column) operand reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2);} } while (0)

// ---- test 1: C[32][32] = A[32][16] * B[16][32]
__global__ void t1(const float* A, const float* B, float* C) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[r * 16 + 8 * h + j]; b[j] = (__bf16)B[(8 * h + j) * 32 + r]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int g = 0; g < 16; ++g) { int row = (g & 3) + 8 * (g >> 2) + 4 * h; C[row * 32 + r] = c[g]; }
}

// ---- test 2: X[32][32] = A1[32][16]*B1[16][32];  Y[32][32] = A2[32][32] * X  (sum over X's row index)
//      X regs 8s..8s+7 -> bf16 fragment of k-step s; element j of lane-half h is row 16s+8(j>>2)+4h+(j&3)
__global__ void t2(const float* A1, const float* B1, const float* A2, float* Y) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A1[r * 16 + 8 * h + j]; b[j] = (__bf16)B1[(8 * h + j) * 32 + r]; }
  f32x16 x = {0};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
  f32x16 y = {0};
  for (int s = 0; s < 2; ++s) {
    bf16x8 xb, a2;
    for (int j = 0; j < 8; ++j) {
      xb[j] = (__bf16)x[8 * s + j];
      int krow = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      a2[j] = (__bf16)A2[r * 32 + krow];
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0, 0, 0);
  }
  for (int g = 0; g < 16; ++g) { int row = (g & 3) + 8 * (g >> 2) + 4 * h; Y[row * 32 + r] = y[g]; }
}

// ---- test 3: ds_read_b64_tr_b16.  LDS holds M[rows][64 cols] of 16-bit ints (row-major, 128-B rows).
// Per 16-lane group g: lane 4q+p supplies &M[r0+q][c0+4p]; lane i receives M[r0..r0+3][c0+i].
__global__ void t3(const short* M, short* out /*[64][4]*/, int r0, int c0) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = M[i];
  __syncthreads();
  int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  // group g reads block rows r0+4g.. , cols c0..c0+15
  const short* addr = &lds[(r0 + 4 * g + q) * 64 + c0 + 4 * p];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}

int main() {
  int fails = 0;
  // test 1
  {
    std::vector<float> A(32 * 16), B(16 * 32), C(32 * 32), R(32 * 32, 0.f);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 3 + k * 5) % 7 - 3);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 2 + j * 11 + 1) % 9 - 4);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j]; R[i * 32 + j] = s; }
    float *dA, *dB, *dC; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    t1<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) if (C[i] != R[i]) ++bad;
    printf("test1 mfma_32x32x16 maps: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  // test 2
  {
    std::vector<float> A1(32 * 16), B1(16 * 32), A2(32 * 32), X(32 * 32), Y(32 * 32), R(32 * 32);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A1[i * 16 + k] = (float)((i * 3 + k * 5) % 5 - 2);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B1[k * 32 + j] = (float)((k * 2 + j * 7 + 1) % 3 - 1);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 32; ++k) A2[i * 32 + k] = (float)((i * 5 + k * 3 + 2) % 5 - 2);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A1[i * 16 + k] * B1[k * 32 + j]; X[i * 32 + j] = s; }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += A2[i * 32 + k] * X[k * 32 + j]; R[i * 32 + j] = s; }
    float *dA1, *dB1, *dA2, *dY; CK(hipMalloc(&dA1, A1.size() * 4)); CK(hipMalloc(&dB1, B1.size() * 4)); CK(hipMalloc(&dA2, A2.size() * 4)); CK(hipMalloc(&dY, Y.size() * 4));
    CK(hipMemcpy(dA1, A1.data(), A1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB1, B1.data(), B1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dA2, A2.data(), A2.size() * 4, hipMemcpyHostToDevice));
    t2<<<1, 64>>>(dA1, dB1, dA2, dY); CK(hipDeviceSynchronize());
    CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) if (Y[i] != R[i]) ++bad;
    printf("test2 acc-as-B-operand (A*X): %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  // test 3
  {
    std::vector<short> M(64 * 64), out(64 * 4);
    for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c) M[r * 64 + c] = (short)(r * 100 + c);
    short *dM, *dO; CK(hipMalloc(&dM, M.size() * 2)); CK(hipMalloc(&dO, out.size() * 2));
    CK(hipMemcpy(dM, M.data(), M.size() * 2, hipMemcpyHostToDevice));
    int r0 = 8, c0 = 16;
    t3<<<1, 64>>>(dM, dO, r0, c0); CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dO, out.size() * 2, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; for (int e = 0; e < 4; ++e) { short exp = (short)((r0 + 4 * g + e) * 100 + c0 + i); if (out[l * 4 + e] != exp) { if (bad < 8) printf("  lane %d e %d got %d exp %d\n", l, e, out[l * 4 + e], exp); ++bad; } } }
    printf("test3 ds_read_b64_tr_b16: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  printf(fails ? "PROBE FAIL\n" : "PROBE OK\n");
  return fails ? 1 : 0;
}
