// Fused log-softmax statistics over vocabulary rows for gfx950 — HBM-bound, one pass per direction.
//   fwd: per row r of logits[R, V] (bf16/f16): lse[r] = ln Σ exp(x/T), ent[r] = lse − Σ p·x/T,
//        lp[r] = x[label[r]]/T − lse[r]                       (vocab_parallel.py:13-27 arithmetic, fp32)
//   bwd: logits are overwritten IN PLACE by d(loss)/d(logits):
//        g[r,j] = ( p_j·(−G[r] + ge[r]·(lse[r] − ent[r] − x_j/T)) + glp[r]·[j == label[r]] ) / T
//        where G[r] = glp[r] + gextra[r] is the summed gradient of every log-prob picked from row r
//        (a fork node has one picked token per child; the extra one-hot terms are added by the caller).
// One 256-thread workgroup per row, 16-byte loads, online (max, Σexp, Σexp·x) per lane, block reduce.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dta.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <int DT> struct LTy;
template <> struct LTy<DTA_BF16> { using e = __bf16; using v8 = bf16x8; };
template <> struct LTy<DTA_F16> { using e = _Float16; using v8 = f16x8; };

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct Stat { float m, s, t; };   // running max (log2 domain of scaled x), Σ 2^(y−m), Σ 2^(y−m)·y   with y = x·LOG2E/T

__device__ __forceinline__ Stat merge(Stat a, Stat b) {
  const float m = fmaxf(a.m, b.m);
  const float fa = __builtin_amdgcn_exp2f(a.m - m), fb = __builtin_amdgcn_exp2f(b.m - m);
  return Stat{m, a.s * fa + b.s * fb, a.t * fa + b.t * fb};
}

__device__ __forceinline__ Stat block_reduce(Stat v, Stat* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Stat w{__shfl_xor(v.m, o), __shfl_xor(v.s, o), __shfl_xor(v.t, o)};
    v = merge(v, w);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  Stat r = sh[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = merge(r, sh[w]);
  return r;
}

template <int DT>
__global__ __launch_bounds__(256) void logprob_entropy_fwd_kernel(const void* __restrict__ logits_, const int64_t* __restrict__ labels,
                                                                  float* __restrict__ lse, float* __restrict__ ent, float* __restrict__ lp,
                                                                  float* __restrict__ stats, int R, int V, int64_t stride, float inv_temp) {
  using e = typename LTy<DT>::e; using v8 = typename LTy<DT>::v8;
  __shared__ Stat sh[4];
  const int row = blockIdx.x;
  const e* x = reinterpret_cast<const e*>(logits_) + (int64_t)row * stride;
  const float k = LOG2E * inv_temp;
  Stat st{-1e30f, 0.f, 0.f};
  const int nv = V >> 3;
  for (int i = threadIdx.x; i < nv; i += 256) {
    const v8 v = *reinterpret_cast<const v8*>(x + 8 * i);
    float y[8]; float mx = -1e30f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { y[j] = (float)v[j] * k; mx = fmaxf(mx, y[j]); }
    const float m = fmaxf(st.m, mx);
    const float f = __builtin_amdgcn_exp2f(st.m - m);
    float s = st.s * f, t = st.t * f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float p = __builtin_amdgcn_exp2f(y[j] - m); s += p; t = __builtin_fmaf(p, y[j], t); }
    st = Stat{m, s, t};
  }
  for (int i = (nv << 3) + threadIdx.x; i < V; i += 256) {            // tail when V % 8 != 0
    const float y = (float)x[i] * k;
    const float m = fmaxf(st.m, y); const float f = __builtin_amdgcn_exp2f(st.m - m); const float p = __builtin_amdgcn_exp2f(y - m);
    st = Stat{m, st.s * f + p, __builtin_fmaf(p, y, st.t * f)};
  }
  st = block_reduce(st, sh);
  if (threadIdx.x == 0 && stats) {
    // vocab-sharded use: raw per-shard statistics (log2 domain of the scaled logits) for a cross-rank combine
    const int64_t lab = labels ? labels[row] : -1;
    stats[4 * row] = st.m; stats[4 * row + 1] = st.s; stats[4 * row + 2] = st.t;
    stats[4 * row + 3] = (lab >= 0 && lab < V) ? (float)x[lab] * inv_temp : 0.f;
  } else if (threadIdx.x == 0) {
    const float lse2 = st.m + __builtin_amdgcn_logf(st.s);               // log2 domain
    const float l = lse2 * LN2;
    lse[row] = l;
    if (ent) ent[row] = l - (st.t / st.s) * LN2;                         // H = lse − E[x/T]
    if (lp) { const int64_t lab = labels[row]; lp[row] = (lab >= 0 && lab < V) ? (float)x[lab] * inv_temp - l : 0.f; }
  }
}

template <int DT>
__global__ __launch_bounds__(256) void logprob_entropy_bwd_kernel(void* __restrict__ logits_, const int64_t* __restrict__ labels,
                                                                  const float* __restrict__ lse, const float* __restrict__ ent,
                                                                  const float* __restrict__ glp, const float* __restrict__ gextra,
                                                                  const float* __restrict__ gent,
                                                                  int R, int V, int64_t stride, float inv_temp) {
  using e = typename LTy<DT>::e; using v8 = typename LTy<DT>::v8;
  const int row = blockIdx.x;
  e* x = reinterpret_cast<e*>(logits_) + (int64_t)row * stride;
  const float l = lse[row];
  const float ge = gent ? gent[row] : 0.f;
  const float g1 = glp ? glp[row] : 0.f;
  const float G = g1 + (gextra ? gextra[row] : 0.f);
  const float a = -G + ge * (l - (ent ? ent[row] : 0.f));
  const int64_t lab = labels ? labels[row] : -1;
  const float k = LOG2E * inv_temp, l2 = l * LOG2E;
  const int nv = V >> 3;
  for (int i = threadIdx.x; i < nv; i += 256) {
    v8 v = *reinterpret_cast<const v8*>(x + 8 * i);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xs = (float)v[j] * inv_temp;
      const float p = __builtin_amdgcn_exp2f(__builtin_fmaf((float)v[j], k, -l2));
      float g = p * (a - ge * xs);
      if (8 * i + j == lab) g += g1;
      v[j] = (e)(g * inv_temp);
    }
    *reinterpret_cast<v8*>(x + 8 * i) = v;
  }
  for (int i = (nv << 3) + threadIdx.x; i < V; i += 256) {
    const float xs = (float)x[i] * inv_temp;
    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf((float)x[i], k, -l2));
    float g = p * (a - ge * xs);
    if (i == lab) g += g1;
    x[i] = (e)(g * inv_temp);
  }
}

}  // namespace

static int logprob_fwd_launch(const void* logits, const int64_t* labels, float* lse, float* entropy, float* logprob, float* stats,
                              int32_t R, int32_t V, int64_t row_stride, float temperature, int32_t dtype, void* stream);

extern "C" int dta_logprob_entropy_fwd(const void* logits, const int64_t* labels, float* lse, float* entropy, float* logprob,
                                       int32_t R, int32_t V, int64_t row_stride, float temperature, int32_t dtype, void* stream) {
  if (!lse) return DTA_EINVAL;
  return logprob_fwd_launch(logits, labels, lse, entropy, logprob, nullptr, R, V, row_stride, temperature, dtype, stream);
}

/* Vocab-sharded variant: stats[r] = {m, s, t, picked} of THIS shard — m = max_j y_j, s = sum 2^(y_j-m),
 * t = sum 2^(y_j-m)*y_j with y = x*log2(e)/T, picked = x[labels[r]]/T if 0 <= labels[r] < V (labels are
 * shard-local, -1 = owned elsewhere) else 0.  Ranks combine them (MAX of m, then SUM of rescaled s, t, picked):
 * the arithmetic of vocab_parallel.py:125-160 / 258-300 with one packed SUM all-reduce. */
extern "C" int dta_logprob_entropy_shard_stats(const void* logits, const int64_t* labels, float* stats,
                                               int32_t R, int32_t V, int64_t row_stride, float temperature, int32_t dtype, void* stream) {
  if (!stats) return DTA_EINVAL;
  return logprob_fwd_launch(logits, labels, nullptr, nullptr, nullptr, stats, R, V, row_stride, temperature, dtype, stream);
}

static int logprob_fwd_launch(const void* logits, const int64_t* labels, float* lse, float* entropy, float* logprob, float* stats,
                              int32_t R, int32_t V, int64_t row_stride, float temperature, int32_t dtype, void* stream) {
  if (!logits || R <= 0 || V <= 0 || (logprob && !labels) || !(temperature > 0.f)) return DTA_EINVAL;
  if (dtype != DTA_BF16 && dtype != DTA_F16) return DTA_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(logits) & 15) || (row_stride % 8)) return DTA_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  if (dtype == DTA_BF16) hipLaunchKernelGGL(logprob_entropy_fwd_kernel<DTA_BF16>, dim3(R), dim3(256), 0, st, logits, labels, lse, entropy, logprob, stats, R, V, row_stride, 1.f / temperature);
  else hipLaunchKernelGGL(logprob_entropy_fwd_kernel<DTA_F16>, dim3(R), dim3(256), 0, st, logits, labels, lse, entropy, logprob, stats, R, V, row_stride, 1.f / temperature);
  return hipGetLastError() == hipSuccess ? DTA_OK : DTA_ELAUNCH;
}

extern "C" int dta_logprob_entropy_bwd(void* logits_inout, const int64_t* labels, const float* lse, const float* entropy,
                                       const float* g_logprob, const float* g_extra, const float* g_entropy,
                                       int32_t R, int32_t V, int64_t row_stride, float temperature, int32_t dtype, void* stream) {
  if (!logits_inout || !lse || R <= 0 || V <= 0 || (g_entropy && !entropy) || (g_logprob && !labels) || !(temperature > 0.f)) return DTA_EINVAL;
  if (dtype != DTA_BF16 && dtype != DTA_F16) return DTA_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(logits_inout) & 15) || (row_stride % 8)) return DTA_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  if (dtype == DTA_BF16) hipLaunchKernelGGL(logprob_entropy_bwd_kernel<DTA_BF16>, dim3(R), dim3(256), 0, st, logits_inout, labels, lse, entropy, g_logprob, g_extra, g_entropy, R, V, row_stride, 1.f / temperature);
  else hipLaunchKernelGGL(logprob_entropy_bwd_kernel<DTA_F16>, dim3(R), dim3(256), 0, st, logits_inout, labels, lse, entropy, g_logprob, g_extra, g_entropy, R, V, row_stride, 1.f / temperature);
  return hipGetLastError() == hipSuccess ? DTA_OK : DTA_ELAUNCH;
}
