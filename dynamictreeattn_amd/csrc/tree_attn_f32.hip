// fp32 tree attention for gfx950 - the CORRECTNESS path of fp32 models (the reference runs `--dtype fp32` through sdpa,
// run.py:122-132; SURVEY §8d recommends an fp32 tree-vs-dense check at <= 1e-5, far below the bf16 noise floor).
//
// Same visibility rule, same buffers and the same workspace conventions as the MFMA kernels of tree_attn.hip
//   packed trie:  key s visible to query t  <=>  s <= t < subtree_end[s]        stack form: subtree_end == NULL, t = q_offset + row
//   lse [Hq, Tq]   = log2-domain log-sum-exp of the scaled scores (m + log2 l with m = max(c·s), c = scale·log2 e)
//   delta [Hq, Tq] = -rowsum(dO ∘ O)
// but plain fp32 FMAs (157 TFLOP/s vector rate; an fp32 MFMA form would buy 2x at most and this is not a performance path):
// a row (query row in fwd / dQ, key row in dK/dV) is owned by FOUR adjacent lanes with 32 of the 128 head dims each, the rows of
// the other side are staged 32 at a time through LDS and broadcast-read; dot products are reduced across the quad with two DPP
// shuffles.  One workgroup = 64 rows x 1 head (fwd, dQ) or 64 keys x 1 kv head over its whole query range and GQA group (dK/dV:
// no atomics, no slabs - the split-Q work units of the MFMA kernel are ignored here).  Deterministic by construction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dta_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int ROWS = 64;        // rows owned by a workgroup (4 lanes each -> 256 threads)
constexpr int ST = 32;          // rows of the other side staged per LDS tile

struct P32 {
  const float *q, *k, *v, *o, *dout;
  float *out, *dq, *dk, *dv, *lse_w, *delta;
  const float* lse_r;
  const int32_t *subtree_end, *run_ptr, *runs, *ktile_qend;
  int32_t Tq, Tk, q_offset, Hq, Hkv, group;
  int64_t q_st, q_sh, kv_st, kv_sh, v_st, v_sh, o_st, o_sh, dq_st, dq_sh, dkv_st, dkv_sh;
  float scale; int32_t accumulate;
};

__device__ __forceinline__ float quad_sum(float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); return v; }

// 32 rows x 128 floats of `base` (row stride `st` elements) starting at row r0, rows clamped to [0, rmax), into img
__device__ __forceinline__ void stage_rows(float* img, const float* base, int64_t st, int r0, int rmax, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i, row = id >> 5, c4 = id & 31;
    int gr = r0 + row; gr = gr < rmax ? gr : rmax - 1; gr = gr < 0 ? 0 : gr;
    reinterpret_cast<float4*>(img)[row * 32 + c4] = *reinterpret_cast<const float4*>(base + (int64_t)gr * st + 4 * c4);
  }
}

__device__ __forceinline__ void load32(float* dst, const float* src) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float4 t = reinterpret_cast<const float4*>(src)[i]; dst[4 * i] = t.x; dst[4 * i + 1] = t.y; dst[4 * i + 2] = t.z; dst[4 * i + 3] = t.w; }
}

// key runs of this workgroup's 64 query rows: the run list of their 128-row query tile (a superset; the per-pair test below is
// the complete visibility rule) or, in the stack form, the single run [0, q_offset + last row + 1)
struct Runs {
  const int32_t* runs; int ri, re, k0, kend;
  __device__ __forceinline__ bool next() {
    while (ri < re) {
      if (runs) { k0 = runs[4 * ri]; kend = runs[4 * ri + 1]; }
      ++ri;
      if (k0 < kend) return true;
    }
    return false;
  }
};
__device__ __forceinline__ Runs runs_of(const P32& p, int q0) {
  Runs r; r.runs = p.runs;
  if (p.runs) { const int qt = q0 / DTA_QTILE; r.ri = p.run_ptr[qt]; r.re = p.run_ptr[qt + 1]; r.k0 = r.kend = 0; }
  else { r.ri = 0; r.re = 1; r.k0 = 0; const int last = p.q_offset + (q0 + ROWS < p.Tq ? q0 + ROWS : p.Tq); r.kend = last < p.Tk ? last : p.Tk; }
  return r;
}

__global__ __launch_bounds__(256) void tree_attn_fwd_f32_kernel(P32 p) {
  __shared__ __attribute__((aligned(16))) float Ks[ST * 128];
  __shared__ __attribute__((aligned(16))) float Vs[ST * 128];
  __shared__ int se_s[ST];
  const int tid = threadIdx.x, row = tid >> 2, part = tid & 3;
  const int hq = blockIdx.y, kvh = hq / p.group;
  const int q0 = blockIdx.x * ROWS, qrow = q0 + row, qrow_c = qrow < p.Tq ? qrow : p.Tq - 1, qidx = p.q_offset + qrow;
  float q[32], o[32];
  load32(q, p.q + (int64_t)qrow_c * p.q_st + (int64_t)hq * p.q_sh + 32 * part);
#pragma unroll
  for (int d = 0; d < 32; ++d) o[d] = 0.f;
  float m = -1e30f, l = 0.f;
  const float c = p.scale * LOG2E;
  const float* kb = p.k + (int64_t)kvh * p.kv_sh; const float* vb = p.v + (int64_t)kvh * p.v_sh;
  Runs rn = runs_of(p, q0);
  while (rn.next()) {
    for (int k0 = rn.k0; k0 < rn.kend; k0 += ST) {
      __syncthreads();
      stage_rows(Ks, kb, p.kv_st, k0, p.Tk, tid); stage_rows(Vs, vb, p.v_st, k0, p.Tk, tid);
      if (tid < ST) { const int ki = k0 + tid; se_s[tid] = (p.subtree_end && ki < p.Tk) ? p.subtree_end[ki] : 0x7fffffff; }
      __syncthreads();
      const int n = rn.kend - k0 < ST ? rn.kend - k0 : ST;
      for (int j = 0; j < n; ++j) {
        const float* kr = Ks + j * 128 + 32 * part;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) s = __builtin_fmaf(q[d], kr[d], s);
        s = quad_sum(s);
        const int key = k0 + j;
        if (key <= qidx && qidx < se_s[j] && key < p.Tk) {
          const float sc = s * c, mn = fmaxf(m, sc);
          const float alpha = __builtin_amdgcn_exp2f(m - mn), pj = __builtin_amdgcn_exp2f(sc - mn);
          const float* vr = Vs + j * 128 + 32 * part;
          l = __builtin_fmaf(l, alpha, pj);
#pragma unroll
          for (int d = 0; d < 32; ++d) o[d] = __builtin_fmaf(pj, vr[d], o[d] * alpha);
          m = mn;
        }
      }
    }
  }
  if (qrow < p.Tq) {
    const float inv = 1.f / l;
    float* op = p.out + (int64_t)qrow * p.o_st + (int64_t)hq * p.o_sh + 32 * part;
#pragma unroll
    for (int i = 0; i < 8; ++i) reinterpret_cast<float4*>(op)[i] = make_float4(o[4 * i] * inv, o[4 * i + 1] * inv, o[4 * i + 2] * inv, o[4 * i + 3] * inv);
    if (part == 0) p.lse_w[(int64_t)hq * p.Tq + qrow] = m + __builtin_amdgcn_logf(l);      // v_log_f32 = log2
  }
}

__global__ __launch_bounds__(256) void tree_attn_bwd_dq_f32_kernel(P32 p) {
  __shared__ __attribute__((aligned(16))) float Ks[ST * 128];
  __shared__ __attribute__((aligned(16))) float Vs[ST * 128];
  __shared__ int se_s[ST];
  const int tid = threadIdx.x, row = tid >> 2, part = tid & 3;
  const int hq = blockIdx.y, kvh = hq / p.group;
  const int q0 = blockIdx.x * ROWS, qrow = q0 + row, qrow_c = qrow < p.Tq ? qrow : p.Tq - 1, qidx = p.q_offset + qrow;
  float q[32], dof[32], dq[32];
  load32(q, p.q + (int64_t)qrow_c * p.q_st + (int64_t)hq * p.q_sh + 32 * part);
  load32(dof, p.dout + (int64_t)qrow_c * p.o_st + (int64_t)hq * p.o_sh + 32 * part);
  float delta = 0.f;
  {
    float of[32];
    load32(of, p.o + (int64_t)qrow_c * p.o_st + (int64_t)hq * p.o_sh + 32 * part);
#pragma unroll
    for (int d = 0; d < 32; ++d) { delta = __builtin_fmaf(dof[d], of[d], delta); dq[d] = 0.f; }
  }
  delta = quad_sum(delta);
  const float lse2 = p.lse_r[(int64_t)hq * p.Tq + qrow_c];
  if (part == 0 && qrow < p.Tq) p.delta[(int64_t)hq * p.Tq + qrow] = -delta;
  const float c = p.scale * LOG2E;
  const float* kb = p.k + (int64_t)kvh * p.kv_sh; const float* vb = p.v + (int64_t)kvh * p.v_sh;
  Runs rn = runs_of(p, q0);
  while (rn.next()) {
    for (int k0 = rn.k0; k0 < rn.kend; k0 += ST) {
      __syncthreads();
      stage_rows(Ks, kb, p.kv_st, k0, p.Tk, tid); stage_rows(Vs, vb, p.v_st, k0, p.Tk, tid);
      if (tid < ST) { const int ki = k0 + tid; se_s[tid] = (p.subtree_end && ki < p.Tk) ? p.subtree_end[ki] : 0x7fffffff; }
      __syncthreads();
      const int n = rn.kend - k0 < ST ? rn.kend - k0 : ST;
      for (int j = 0; j < n; ++j) {
        const float* kr = Ks + j * 128 + 32 * part; const float* vr = Vs + j * 128 + 32 * part;
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) { s = __builtin_fmaf(q[d], kr[d], s); dp = __builtin_fmaf(dof[d], vr[d], dp); }
        s = quad_sum(s); dp = quad_sum(dp);
        const int key = k0 + j;
        if (key <= qidx && qidx < se_s[j] && key < p.Tk) {
          const float ds = __builtin_amdgcn_exp2f(__builtin_fmaf(s, c, -lse2)) * (dp - delta);
#pragma unroll
          for (int d = 0; d < 32; ++d) dq[d] = __builtin_fmaf(ds, kr[d], dq[d]);
        }
      }
    }
  }
  if (qrow < p.Tq) {
    float* dp_ = p.dq + (int64_t)qrow * p.dq_st + (int64_t)hq * p.dq_sh + 32 * part;
    const float sc = p.scale;
#pragma unroll
    for (int i = 0; i < 8; ++i) reinterpret_cast<float4*>(dp_)[i] = make_float4(dq[4 * i] * sc, dq[4 * i + 1] * sc, dq[4 * i + 2] * sc, dq[4 * i + 3] * sc);
  }
}

// key-owned sweep: 64 keys of one kv head x every query that can see them x the heads of the GQA group
__global__ __launch_bounds__(256) void tree_attn_bwd_dkv_f32_kernel(P32 p) {
  __shared__ __attribute__((aligned(16))) float Qs[ST * 128];
  __shared__ __attribute__((aligned(16))) float Ds[ST * 128];
  __shared__ float lse_s[ST], nd_s[ST];
  const int tid = threadIdx.x, row = tid >> 2, part = tid & 3;
  const int kvh = blockIdx.y;
  const int k0 = blockIdx.x * ROWS, key = k0 + row, key_c = key < p.Tk ? key : p.Tk - 1;
  float kf[32], vf[32], dk[32], dv[32];
  load32(kf, p.k + (int64_t)key_c * p.kv_st + (int64_t)kvh * p.kv_sh + 32 * part);
  load32(vf, p.v + (int64_t)key_c * p.v_st + (int64_t)kvh * p.v_sh + 32 * part);
#pragma unroll
  for (int d = 0; d < 32; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
  const int se = (p.subtree_end && key < p.Tk) ? p.subtree_end[key_c] : 0x7fffffff;
  const float c = p.scale * LOG2E;
  // global query indices [t_begin, t_end) that may see a key of this block
  int t_begin = k0 > p.q_offset ? k0 : p.q_offset;
  int t_end = p.q_offset + p.Tq;
  if (p.ktile_qend) {                                  // max subtree_end over each DTA_KTILE-key tile (this block lies inside one)
    const int qe = p.ktile_qend[k0 / DTA_KTILE];
    t_end = qe < t_end ? qe : t_end;
  }
  for (int g = 0; g < p.group; ++g) {
    const int hq = kvh * p.group + g;
    const float* qb = p.q + (int64_t)hq * p.q_sh; const float* dob = p.dout + (int64_t)hq * p.o_sh;
    for (int t0 = t_begin; t0 < t_end; t0 += ST) {
      const int r0 = t0 - p.q_offset;
      __syncthreads();
      stage_rows(Qs, qb, p.q_st, r0, p.Tq, tid); stage_rows(Ds, dob, p.o_st, r0, p.Tq, tid);
      if (tid < ST) { int rr = r0 + tid; rr = rr < p.Tq ? rr : p.Tq - 1; lse_s[tid] = p.lse_r[(int64_t)hq * p.Tq + rr]; nd_s[tid] = p.delta[(int64_t)hq * p.Tq + rr]; }
      __syncthreads();
      const int n = t_end - t0 < ST ? t_end - t0 : ST;
      for (int i = 0; i < n; ++i) {
        const float* qr = Qs + i * 128 + 32 * part; const float* dr = Ds + i * 128 + 32 * part;
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) { s = __builtin_fmaf(kf[d], qr[d], s); dp = __builtin_fmaf(vf[d], dr[d], dp); }
        s = quad_sum(s); dp = quad_sum(dp);
        const int t = t0 + i;
        if (key <= t && t < se && key < p.Tk) {
          const float pj = __builtin_amdgcn_exp2f(__builtin_fmaf(s, c, -lse_s[i]));
          const float ds = pj * (dp + nd_s[i]);
#pragma unroll
          for (int d = 0; d < 32; ++d) { dv[d] = __builtin_fmaf(pj, dr[d], dv[d]); dk[d] = __builtin_fmaf(ds, qr[d], dk[d]); }
        }
      }
    }
  }
  if (key < p.Tk) {
    float* dkp = p.dk + (int64_t)key * p.dkv_st + (int64_t)kvh * p.dkv_sh + 32 * part;
    float* dvp = p.dv + (int64_t)key * p.dkv_st + (int64_t)kvh * p.dkv_sh + 32 * part;
    const float sc = p.scale;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float4 a = make_float4(dk[4 * i] * sc, dk[4 * i + 1] * sc, dk[4 * i + 2] * sc, dk[4 * i + 3] * sc);
      float4 b = make_float4(dv[4 * i], dv[4 * i + 1], dv[4 * i + 2], dv[4 * i + 3]);
      if (p.accumulate) {                               // 1: add to the caller's gradient, 2: add into the fp32 grad-KV stacks - the same thing in fp32
        const float4 x = reinterpret_cast<const float4*>(dkp)[i], y = reinterpret_cast<const float4*>(dvp)[i];
        a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
      }
      reinterpret_cast<float4*>(dkp)[i] = a; reinterpret_cast<float4*>(dvp)[i] = b;
    }
  }
}

}  // namespace

int dta_attn_fwd_f32(const void* q, const void* k, const void* v, void* out, float* lse, const int32_t* subtree_end, const int32_t* run_ptr,
                     const int32_t* runs, int32_t Tq, int32_t Tk, int32_t q_offset, int32_t Hq, int32_t Hkv,
                     int64_t q_st, int64_t q_sh, int64_t kv_st, int64_t kv_sh, int64_t v_st, int64_t v_sh, int64_t o_st, int64_t o_sh,
                     float scale, hipStream_t st) {
  P32 p{};
  p.q = (const float*)q; p.k = (const float*)k; p.v = (const float*)v; p.out = (float*)out; p.lse_w = lse;
  p.subtree_end = subtree_end; p.run_ptr = run_ptr; p.runs = runs;
  p.Tq = Tq; p.Tk = Tk; p.q_offset = q_offset; p.Hq = Hq; p.Hkv = Hkv; p.group = Hq / Hkv;
  p.q_st = q_st; p.q_sh = q_sh; p.kv_st = kv_st; p.kv_sh = kv_sh; p.v_st = v_st; p.v_sh = v_sh; p.o_st = o_st; p.o_sh = o_sh; p.scale = scale;
  hipLaunchKernelGGL(tree_attn_fwd_f32_kernel, dim3((Tq + ROWS - 1) / ROWS, Hq), dim3(256), 0, st, p);
  return DTA_LAUNCH_STATUS();
}

int dta_attn_bwd_f32(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, float* delta,
                     void* dq, void* dk, void* dv, const int32_t* subtree_end, const int32_t* run_ptr, const int32_t* runs, const int32_t* ktile_qend,
                     int32_t Tq, int32_t Tk, int32_t q_offset, int32_t Hq, int32_t Hkv,
                     int64_t q_st, int64_t q_sh, int64_t kv_st, int64_t kv_sh, int64_t v_st, int64_t v_sh, int64_t o_st, int64_t o_sh,
                     int64_t dq_st, int64_t dq_sh, int64_t dkv_st, int64_t dkv_sh, float scale, int32_t accumulate, int32_t which, hipStream_t st) {
  P32 p{};
  p.q = (const float*)q; p.k = (const float*)k; p.v = (const float*)v; p.o = (const float*)out; p.dout = (const float*)dout;
  p.lse_r = lse; p.delta = delta; p.dq = (float*)dq; p.dk = (float*)dk; p.dv = (float*)dv;
  p.subtree_end = subtree_end; p.run_ptr = run_ptr; p.runs = runs; p.ktile_qend = ktile_qend;
  p.Tq = Tq; p.Tk = Tk; p.q_offset = q_offset; p.Hq = Hq; p.Hkv = Hkv; p.group = Hq / Hkv;
  p.q_st = q_st; p.q_sh = q_sh; p.kv_st = kv_st; p.kv_sh = kv_sh; p.v_st = v_st; p.v_sh = v_sh; p.o_st = o_st; p.o_sh = o_sh;
  p.dq_st = dq_st; p.dq_sh = dq_sh; p.dkv_st = dkv_st; p.dkv_sh = dkv_sh; p.scale = scale; p.accumulate = accumulate;
  if (which & 1) hipLaunchKernelGGL(tree_attn_bwd_dq_f32_kernel, dim3((Tq + ROWS - 1) / ROWS, Hq), dim3(256), 0, st, p);      // also writes -delta
  if (which & 2) hipLaunchKernelGGL(tree_attn_bwd_dkv_f32_kernel, dim3((Tk + ROWS - 1) / ROWS, Hkv), dim3(256), 0, st, p);
  return DTA_LAUNCH_STATUS();                           // (which & 4, the slab finalize of the MFMA path, has nothing to do here)
}
