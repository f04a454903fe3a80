// Fused log-softmax statistics over vocabulary rows for gfx950 — HBM-bound, one pass per direction.
//   fwd: per row r of logits[R, V] (bf16 / f16 / f32): lse[r] = ln Σ exp(x/T), ent[r] = lse − Σ p·x/T,
//        lp[r] = x[label[r]]/T − lse[r]                       (vocab_parallel.py:13-27 arithmetic, fp32)
//        and, for every EXTRA label e of the row (CSR extra_ptr/extra_labels: a fork node of the trie predicts one token per
//        child, tree_training_engine.py:205-209, 217-220, 369-372), extra_lp[e] = x[extra_labels[e]]/T − lse[r].
//   bwd: dLoss/dlogits, written to `out` (out == logits: in place):
//        g[r,j] = ( p_j·(−G[r] + ge[r]·(lse[r] − ent[r] − x_j/T)) + Σ_{picked j} g_picked ) / T
//        where G[r] = glp[r] + Σ_e g_extra_lp[e] is the summed gradient of every log-prob picked from row r.
// One 256-thread workgroup per row, 16-byte loads, online (max, Σexp, Σexp·x) per lane, block reduce.
// Algorithmic HBM bytes: fwd V·sizeof(e) per row (one read); bwd 2·V·sizeof(e) per row (one read, one write).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dta_common.h"

// The [rows, V] logits are 8.6 GB at the bench's shape: each byte is read once here (and once more by the gradient GEMMs, long after it
// has left every cache).  Default: non-temporal accesses (no cache allocation) - forward 1.389 -> 1.240 ms (6.9 TB/s), backward 3.36 -> 3.31 ms at
// [28 160, 151 936] bf16; -DDTA_LOGPROB_NT=0: plain.
#ifndef DTA_LOGPROB_NT
#define DTA_LOGPROB_NT 1
#endif
#if DTA_LOGPROB_NT
#define DTA_LP_LOAD(P) __builtin_nontemporal_load(P)
#define DTA_LP_STORE(P, V) __builtin_nontemporal_store(V, P)
#else
#define DTA_LP_LOAD(P) (*(P))
#define DTA_LP_STORE(P, V) (*(P) = (V))
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) float f32x8;
template <int DT> struct LTy;
template <> struct LTy<DTA_BF16> { using e = __bf16; using v8 = bf16x8; };
template <> struct LTy<DTA_F16> { using e = _Float16; using v8 = f16x8; };
template <> struct LTy<DTA_F32> { using e = float; using v8 = f32x8; };

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

struct FwdArgs {
  const void* logits; const int64_t* labels; const int32_t* extra_ptr; const int64_t* extra_labels;
  float *lse, *ent, *lp, *extra_lp, *stats;
  int R, V; int64_t stride; float inv_temp;
};

template <int DT>
__global__ __launch_bounds__(256) void logprob_entropy_fwd_kernel(FwdArgs a) {
  using e = typename LTy<DT>::e; using v8 = typename LTy<DT>::v8;
  __shared__ Stat sh[4];
  const int row = blockIdx.x, V = a.V;
  const e* x = reinterpret_cast<const e*>(a.logits) + (int64_t)row * a.stride;
  const float k = LOG2E * a.inv_temp;
  Stat st{-1e30f, 0.f, 0.f};
  const int nv = V >> 3;
  for (int i = threadIdx.x; i < nv; i += 256) {
    const v8 v = DTA_LP_LOAD(reinterpret_cast<const v8*>(x + 8 * i));
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
  const int e0 = a.extra_ptr ? a.extra_ptr[row] : 0, e1 = a.extra_ptr ? a.extra_ptr[row + 1] : 0;
  if (a.stats) {
    // vocab-sharded use: raw per-shard statistics (log2 domain of the scaled logits) for a cross-rank combine; picked
    // values are the raw x/T of the labels this shard owns (0 otherwise)
    if (threadIdx.x == 0) {
      const int64_t lab = a.labels ? a.labels[row] : -1;
      a.stats[4 * row] = st.m; a.stats[4 * row + 1] = st.s; a.stats[4 * row + 2] = st.t;
      a.stats[4 * row + 3] = (lab >= 0 && lab < V) ? (float)x[lab] * a.inv_temp : 0.f;
    }
    for (int f = e0 + threadIdx.x; f < e1; f += 256) { const int64_t lab = a.extra_labels[f]; a.extra_lp[f] = (lab >= 0 && lab < V) ? (float)x[lab] * a.inv_temp : 0.f; }
  } else {
    const float l = (st.m + __builtin_amdgcn_logf(st.s)) * LN2;          // v_log_f32 = log2
    if (threadIdx.x == 0) {
      a.lse[row] = l;
      if (a.ent) a.ent[row] = l - (st.t / st.s) * LN2;                    // H = lse − E[x/T]
      if (a.lp) { const int64_t lab = a.labels[row]; a.lp[row] = (lab >= 0 && lab < V) ? (float)x[lab] * a.inv_temp - l : 0.f; }
    }
    for (int f = e0 + threadIdx.x; f < e1; f += 256) { const int64_t lab = a.extra_labels[f]; a.extra_lp[f] = (lab >= 0 && lab < V) ? (float)x[lab] * a.inv_temp - l : 0.f; }
  }
}

struct BwdArgs {
  const void* logits; void* out; const int64_t* labels; const int32_t* extra_ptr; const int64_t* extra_labels;
  const float *lse, *ent, *glp, *gextra, *gent;
  int R, V; int64_t stride, out_stride; float inv_temp;
};

template <int DT>
__global__ __launch_bounds__(256) void logprob_entropy_bwd_kernel(BwdArgs b) {
  using e = typename LTy<DT>::e; using v8 = typename LTy<DT>::v8;
  const int row = blockIdx.x, V = b.V;
  const e* x = reinterpret_cast<const e*>(b.logits) + (int64_t)row * b.stride;
  e* o = reinterpret_cast<e*>(b.out) + (int64_t)row * b.out_stride;
  const float inv_temp = b.inv_temp;
  const float l = b.lse[row];
  const float ge = b.gent ? b.gent[row] : 0.f;
  const float g1 = b.glp ? b.glp[row] : 0.f;
  const int e0 = b.extra_ptr ? b.extra_ptr[row] : 0, e1 = b.extra_ptr ? b.extra_ptr[row + 1] : 0;
  float G = g1;
  for (int f = e0; f < e1; ++f) G += b.gextra[f];                    // a handful per fork row, none elsewhere (uniform loop)
  const float a = -G + ge * (l - (b.ent ? b.ent[row] : 0.f));
  const int64_t lab = b.labels ? b.labels[row] : -1;
  const float k = LOG2E * inv_temp, l2 = l * LOG2E;
  const int nv = V >> 3;
  // d/dx = (p * (a - ge * x/T) + [label] g1) / T  =  p * (c1 + c2 * x) + [label] g1 / T: two instructions per element besides the exponential's two,
  // and the label test once per 16-byte group (it was a compare and a select per element)
  const float c1 = a * inv_temp, c2 = -ge * inv_temp * inv_temp, gl = g1 * inv_temp;
  const int lab8 = lab >= 0 && lab < ((int64_t)nv << 3) ? (int)(lab >> 3) : -1, labj = (int)(lab & 7);
  for (int i = threadIdx.x; i < nv; i += 256) {
    v8 v = DTA_LP_LOAD(reinterpret_cast<const v8*>(x + 8 * i));
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xf = (float)v[j];
      const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(xf, k, -l2));
      g[j] = p * __builtin_fmaf(xf, c2, c1);
    }
    if (i == lab8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] += j == labj ? gl : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (e)g[j];
    DTA_LP_STORE(reinterpret_cast<v8*>(o + 8 * i), v);
  }
  for (int i = (nv << 3) + threadIdx.x; i < V; i += 256) {
    const float xs = (float)x[i] * inv_temp;
    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf((float)x[i], k, -l2));
    float g = p * (a - ge * xs);
    if (i == lab) g += g1;
    o[i] = (e)(g * inv_temp);
  }
  if (e1 > e0) {                                                      // one-hot terms of the extra picks (distinct tokens: children of one node)
    __syncthreads();
    for (int f = e0 + threadIdx.x; f < e1; f += 256) {
      const int64_t le = b.extra_labels[f];
      if (le >= 0 && le < V) o[le] = (e)((float)o[le] + b.gextra[f] * inv_temp);
    }
  }
}

int fwd_launch(FwdArgs a, int32_t dtype, float temperature, void* stream) {
  if (!a.logits || a.R <= 0 || a.V <= 0 || (a.lp && !a.labels) || !(temperature > 0.f)) return DTA_EINVAL;
  if (a.extra_ptr && (!a.extra_labels || !a.extra_lp)) return DTA_EINVAL;
  if (dtype != DTA_BF16 && dtype != DTA_F16 && dtype != DTA_F32) return DTA_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.logits) & (dtype == DTA_F32 ? 31 : 15)) || (a.stride % 8)) return DTA_EALIGN;   // 8-element vector loads
  a.inv_temp = 1.f / temperature;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DTA_REFUSE_IF_PRIOR_ERROR();
  if (dtype == DTA_BF16) hipLaunchKernelGGL(logprob_entropy_fwd_kernel<DTA_BF16>, dim3(a.R), dim3(256), 0, st, a);
  else if (dtype == DTA_F16) hipLaunchKernelGGL(logprob_entropy_fwd_kernel<DTA_F16>, dim3(a.R), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(logprob_entropy_fwd_kernel<DTA_F32>, dim3(a.R), dim3(256), 0, st, a);
  return DTA_LAUNCH_STATUS();
}

}  // namespace

extern "C" int dta_logprob_entropy_fwd(const void* logits, const int64_t* labels, const int32_t* extra_ptr, const int64_t* extra_labels,
                                       float* lse, float* entropy, float* logprob, float* extra_logprob,
                                       int32_t R, int32_t V, int64_t row_stride, float temperature, int32_t dtype, void* stream) {
  if (!lse) return DTA_EINVAL;
  FwdArgs a{logits, labels, extra_ptr, extra_labels, lse, entropy, logprob, extra_logprob, nullptr, R, V, row_stride, 1.f};
  return fwd_launch(a, dtype, temperature, stream);
}

extern "C" int dta_logprob_entropy_shard_stats(const void* logits, const int64_t* labels, const int32_t* extra_ptr, const int64_t* extra_labels,
                                               float* stats, float* extra_picked,
                                               int32_t R, int32_t V, int64_t row_stride, float temperature, int32_t dtype, void* stream) {
  if (!stats) return DTA_EINVAL;
  FwdArgs a{logits, labels, extra_ptr, extra_labels, nullptr, nullptr, nullptr, extra_picked, stats, R, V, row_stride, 1.f};
  return fwd_launch(a, dtype, temperature, stream);
}

extern "C" int dta_logprob_entropy_bwd(const void* logits, void* dlogits, const int64_t* labels, const int32_t* extra_ptr, const int64_t* extra_labels,
                                       const float* lse, const float* entropy,
                                       const float* g_logprob, const float* g_extra_logprob, const float* g_entropy,
                                       int32_t R, int32_t V, int64_t row_stride, int64_t out_row_stride, float temperature, int32_t dtype, void* stream) {
  if (!logits || !dlogits || !lse || R <= 0 || V <= 0 || (g_entropy && !entropy) || (g_logprob && !labels) || !(temperature > 0.f)) return DTA_EINVAL;
  if (extra_ptr && (!extra_labels || !g_extra_logprob)) return DTA_EINVAL;
  if (dtype != DTA_BF16 && dtype != DTA_F16 && dtype != DTA_F32) return DTA_EUNSUPPORTED;
  const uintptr_t am = dtype == DTA_F32 ? 31 : 15;
  if ((reinterpret_cast<uintptr_t>(logits) & am) || (reinterpret_cast<uintptr_t>(dlogits) & am) || (row_stride % 8) || (out_row_stride % 8)) return DTA_EALIGN;
  BwdArgs b{logits, dlogits, labels, extra_ptr, extra_labels, lse, entropy, g_logprob, g_extra_logprob, g_entropy, R, V, row_stride, out_row_stride, 1.f / temperature};
  hipStream_t st = static_cast<hipStream_t>(stream);
  DTA_REFUSE_IF_PRIOR_ERROR();
  if (dtype == DTA_BF16) hipLaunchKernelGGL(logprob_entropy_bwd_kernel<DTA_BF16>, dim3(R), dim3(256), 0, st, b);
  else if (dtype == DTA_F16) hipLaunchKernelGGL(logprob_entropy_bwd_kernel<DTA_F16>, dim3(R), dim3(256), 0, st, b);
  else hipLaunchKernelGGL(logprob_entropy_bwd_kernel<DTA_F32>, dim3(R), dim3(256), 0, st, b);
  return DTA_LAUNCH_STATUS();
}
