// Fused row kernels of the decoder layer for gfx950 — HBM-bound, 16-byte accesses, fp32 math.
//   dta_rmsnorm_fwd/bwd        y = w · cast(x · rsqrt(mean(x²)+eps))                 (Qwen3RMSNorm arithmetic)
//   dta_qk_norm_rope_fwd/bwd   per (token, head) of 128: optional RMSNorm, then RoPE at position = trie depth
//   dta_swiglu_fwd/bwd         y = cast(silu(g)) · u
// These replace ~40 torch elementwise launches per layer (fp32 up-casts included); together they are ~18 ms of a
// 250 ms step (profiles/r1_bench_kernel_stats.csv).  Reference call sites: the model call of
// tree_training_engine.py:182-186, 248-252, 351-353 (third-party transformers Qwen3 layers).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dta_common.h"

// Saved activations read back in a backward kernel (written a whole forward ago: in no cache) are read non-temporally - swiglu_bwd 150.5 ->
// 140.2 us (6.2 TB/s), qk_norm_rope_bwd 53.7 -> 51.6, rmsnorm_bwd 45.9 -> 45.4 at 28 160 rows; -DDTA_EW_NT=0: plain loads.
#ifndef DTA_EW_NT
#define DTA_EW_NT 1
#endif
#if DTA_EW_NT
#define DTA_SAVED_LOAD(P) __builtin_nontemporal_load(P)
#else
#define DTA_SAVED_LOAD(P) (*(P))
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <int DT> struct ETy;
template <> struct ETy<DTA_BF16> { using e = __bf16; using v8 = bf16x8; };
template <> struct ETy<DTA_F16> { using e = _Float16; using v8 = f16x8; };
// fp32 models (the reference's --dtype fp32, run.py:122-132): same kernels, 8 floats = two 16-byte accesses per lane; the
// roundings to the storage type `(e)(...)` are then the identity
typedef float f32x8 __attribute__((ext_vector_type(8), aligned(16)));
template <> struct ETy<DTA_F32> { using e = float; using v8 = f32x8; };
inline bool row_dtype_ok(int dtype) { return dtype == DTA_BF16 || dtype == DTA_F16 || dtype == DTA_F32; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------------------------
// RMSNorm over rows of H (H % 8 == 0; forward: any H, backward: H <= 8192).  One wave per row, 4 rows per workgroup, grid-stride.
// ---------------------------------------------------------------------------------------------
template <int DT, int NA>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const void* __restrict__ x_, const void* __restrict__ delta_, const void* __restrict__ w_,
                                                          void* __restrict__ xout_, void* __restrict__ y_,
                                                          float* __restrict__ rstd, int R, int H, float eps) {
  // NA > 0: the row (H <= 512*NA elements) stays in registers between the sum-of-squares pass and the scaling pass - ONE read of x (and of
  // delta) per row; NA == 0: any H, second pass re-reads the row (L2-hot).
  using e = typename ETy<DT>::e; using v8 = typename ETy<DT>::v8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const e* w = reinterpret_cast<const e*>(w_);
  const int nv = H >> 3;
  for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
    const e* x = reinterpret_cast<const e*>(x_) + (int64_t)row * H;
    e* y = reinterpret_cast<e*>(y_) + (int64_t)row * H;
    float ss = 0.f;
    if constexpr (NA > 0) {
      v8 keep[NA];
      const e* dl = delta_ ? reinterpret_cast<const e*>(delta_) + (int64_t)row * H : nullptr;
      e* xo = delta_ ? reinterpret_cast<e*>(xout_) + (int64_t)row * H : nullptr;
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int i = lane + 64 * a;
        if (i < nv) {
          v8 v = *reinterpret_cast<const v8*>(x + 8 * i);
          if (dl) {
            const v8 d = *reinterpret_cast<const v8*>(dl + 8 * i);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (e)((float)v[j] + (float)d[j]);
            *reinterpret_cast<v8*>(xo + 8 * i) = v;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; ss = __builtin_fmaf(f, f, ss); }
          keep[a] = v;
        }
      }
      ss = wave_sum(ss);
      const float r = __builtin_amdgcn_rsqf(ss / (float)H + eps);
      if (lane == 0) rstd[row] = r;
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int i = lane + 64 * a;
        if (i < nv) {
          const v8 wv = *reinterpret_cast<const v8*>(w + 8 * i);
          v8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) { const e t = (e)((float)keep[a][j] * r); o[j] = (e)((float)wv[j] * (float)t); }
          *reinterpret_cast<v8*>(y + 8 * i) = o;
        }
      }
    } else {
      if (delta_) {                                      // residual stream update fused in: x_out = x + delta (rounded), then normalised
        const e* dl = reinterpret_cast<const e*>(delta_) + (int64_t)row * H;
        e* xo = reinterpret_cast<e*>(xout_) + (int64_t)row * H;
        for (int i = lane; i < nv; i += 64) {
          const v8 v = *reinterpret_cast<const v8*>(x + 8 * i); const v8 d = *reinterpret_cast<const v8*>(dl + 8 * i);
          v8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) { o[j] = (e)((float)v[j] + (float)d[j]); const float f = (float)o[j]; ss = __builtin_fmaf(f, f, ss); }
          *reinterpret_cast<v8*>(xo + 8 * i) = o;
        }
        x = xo;                                          // second pass re-reads the wave's own (L1/L2-hot) row
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      } else {
        for (int i = lane; i < nv; i += 64) {
          const v8 v = *reinterpret_cast<const v8*>(x + 8 * i);
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; ss = __builtin_fmaf(f, f, ss); }
        }
      }
      ss = wave_sum(ss);
      const float r = __builtin_amdgcn_rsqf(ss / (float)H + eps);
      if (lane == 0) rstd[row] = r;
      for (int i = lane; i < nv; i += 64) {
        const v8 v = *reinterpret_cast<const v8*>(x + 8 * i);
        const v8 wv = *reinterpret_cast<const v8*>(w + 8 * i);
        v8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const e t = (e)((float)v[j] * r); o[j] = (e)((float)wv[j] * (float)t); }
        *reinterpret_cast<v8*>(y + 8 * i) = o;
      }
    }
  }
}

// dx = r·(dt − t̂·mean(dt·t̂)), dt = dy·w, t̂ = x·r ;  dw partial per workgroup: Σ_rows dy·t̂.
// NA = v8 groups per lane (dw accumulators in registers): 2/4/8 for H <= 1024/2048/4096, 16 for H <= 8192 (Qwen3-14B/32B hidden 5120).
template <int DT, int NA>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const void* __restrict__ x_, const void* __restrict__ w_, const void* __restrict__ dy_,
                                                          const void* __restrict__ dres_,
                                                          const float* __restrict__ rstd, void* __restrict__ dx_, float* __restrict__ dw_part,
                                                          int R, int H) {
  using e = typename ETy<DT>::e; using v8 = typename ETy<DT>::v8;
  __shared__ float red[4 * 64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const e* w = reinterpret_cast<const e*>(w_);
  const int nv = H >> 3;
  const int per_lane = (nv + 63) >> 6;                      // <= NA
  constexpr bool KEEP = NA * sizeof(v8) <= 128;             // x and dy of the row stay in registers between the two passes (<= 64 VGPRs)
  float acc[NA][8];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[a][j] = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
    const e* x = reinterpret_cast<const e*>(x_) + (int64_t)row * H;
    const e* dy = reinterpret_cast<const e*>(dy_) + (int64_t)row * H;
    e* dx = reinterpret_cast<e*>(dx_) + (int64_t)row * H;
    const float r = rstd[row];
    float dot = 0.f;
    v8 kx[KEEP ? NA : 1], kg[KEEP ? NA : 1];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int i = lane + 64 * a;
      if (a < per_lane && i < nv) {
        const v8 v = DTA_SAVED_LOAD(reinterpret_cast<const v8*>(x + 8 * i)); const v8 g = *reinterpret_cast<const v8*>(dy + 8 * i);
        const v8 wv = *reinterpret_cast<const v8*>(w + 8 * i);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = (float)v[j] * r; const float gg = (float)g[j]; dot = __builtin_fmaf(gg * (float)wv[j], t, dot); acc[a][j] = __builtin_fmaf(gg, t, acc[a][j]); }
        if constexpr (KEEP) { kx[a] = v; kg[a] = g; }
      }
    }
    dot = wave_sum(dot) / (float)H;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int i = lane + 64 * a;
      if (a < per_lane && i < nv) {
        v8 v, g;
        if constexpr (KEEP) { v = kx[a]; g = kg[a]; }
        else { v = *reinterpret_cast<const v8*>(x + 8 * i); g = *reinterpret_cast<const v8*>(dy + 8 * i); }
        const v8 wv = *reinterpret_cast<const v8*>(w + 8 * i);
        v8 o;
        if (dres_) {                                   // gradient arriving on the residual stream is added here (one pass less)
          const v8 dr = *reinterpret_cast<const v8*>(reinterpret_cast<const e*>(dres_) + (int64_t)row * H + 8 * i);
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float t = (float)v[j] * r; o[j] = (e)(r * ((float)g[j] * (float)wv[j] - t * dot) + (float)dr[j]); }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float t = (float)v[j] * r; o[j] = (e)(r * ((float)g[j] * (float)wv[j] - t * dot)); }
        }
        *reinterpret_cast<v8*>(dx + 8 * i) = o;
      }
    }
  }
  // reduce the 4 waves' dw partials through LDS, one v8-group at a time
  float* out = dw_part + (int64_t)blockIdx.x * H;
#pragma unroll
  for (int a = 0; a < NA; ++a) {                              // static register index; `a < per_lane` is block-uniform
    if (a < per_lane) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) red[(wave * 64 + lane) * 8 + j] = acc[a][j];
      __syncthreads();
      if (wave == 0) {
        const int i = lane + 64 * a;
        if (i < nv) {
#pragma unroll
          for (int j = 0; j < 8; ++j) out[8 * i + j] = red[lane * 8 + j] + red[(64 + lane) * 8 + j] + red[(128 + lane) * 8 + j] + red[(192 + lane) * 8 + j];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// q/k head-norm + RoPE.  x: [T, NH, 128]; 16 lanes own one head (8 elements each); a wave = 4 heads.
// cs: [T, 128] float = {cos[0..63], sin[0..63]} of the token's depth.  w == NULL: RoPE only.
// ---------------------------------------------------------------------------------------------
// HPL = heads of ONE token per 16-lane group (4 when NH % 4 == 0): the token's cos/sin values - 64 bytes per lane, four times the 16 bytes
// of x - are fetched once and reused, and HPL independent 16-byte loads are in flight per lane.
template <int DT, int HPL>
__global__ __launch_bounds__(256) void qk_norm_rope_fwd_kernel(const void* __restrict__ x_, const void* __restrict__ w_, const float* __restrict__ cs,
                                                               void* __restrict__ y_, float* __restrict__ rstd, int64_t n_units, int NH,
                                                               int64_t x_st, float eps) {
  using e = typename ETy<DT>::e; using v8 = typename ETy<DT>::v8;
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const int64_t unit = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);     // global (token, head group) index
  if (unit >= n_units) return;
  const int groups = NH / HPL;
  const int64_t tok = unit / groups; const int head0 = (int)(unit - tok * groups) * HPL;
  const e* x = reinterpret_cast<const e*>(x_) + tok * x_st + (int64_t)head0 * 128 + 8 * sub;
  const int64_t hid0 = tok * NH + head0;
  e* y = reinterpret_cast<e*>(y_) + hid0 * 128 + 8 * sub;
  v8 v[HPL];
#pragma unroll
  for (int h = 0; h < HPL; ++h) v[h] = *reinterpret_cast<const v8*>(x + h * 128);
  // rotate_half partner: element i <-> i +- 64  == lane sub ^ 8 of the same head
  const float* c = cs + tok * 128 + 8 * (sub & 7);
  float cj[8], sj[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { cj[j] = c[j]; sj[j] = sub < 8 ? -c[64 + j] : c[64 + j]; }
  v8 wv;
  if (w_) wv = *reinterpret_cast<const v8*>(reinterpret_cast<const e*>(w_) + 8 * sub);
#pragma unroll
  for (int h = 0; h < HPL; ++h) {
    float a[8];
    if (w_) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)v[h][j]; ss = __builtin_fmaf(f, f, ss); }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
      const float r = __builtin_amdgcn_rsqf(ss * (1.f / 128.f) + eps);
      if (sub == 0) rstd[hid0 + h] = r;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const e t = (e)((float)v[h][j] * r); a[j] = (float)(e)((float)wv[j] * (float)t); }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = (float)v[h][j];
    }
    v8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float other = __shfl_xor(a[j], 8);
      o[j] = (e)(a[j] * cj[j] + other * sj[j]);
    }
    *reinterpret_cast<v8*>(y + h * 128) = o;
  }
}

// dx_ may be dy_ (in place): a lane writes exactly the 16-byte groups it read; its partner's values arrive through registers.
template <int DT, int HPL>
__global__ __launch_bounds__(256) void qk_norm_rope_bwd_kernel(const void* __restrict__ x_, const void* __restrict__ w_, const float* __restrict__ cs,
                                                               const void* dy_, const float* __restrict__ rstd,
                                                               void* dx_, float* __restrict__ dw_part, int64_t n_units, int NH,
                                                               int64_t x_st, int64_t dy_st_t, int64_t dy_st_h, int64_t dx_st) {
  using e = typename ETy<DT>::e; using v8 = typename ETy<DT>::v8;
  __shared__ float red[256 * 8];
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const int groups = NH / HPL;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  v8 wv;
  if (w_) wv = *reinterpret_cast<const v8*>(reinterpret_cast<const e*>(w_) + 8 * sub);
  for (int64_t base = (int64_t)blockIdx.x * 16; base < n_units; base += (int64_t)gridDim.x * 16) {
    const int64_t unit = base + (threadIdx.x >> 6) * 4 + (lane >> 4);
    const bool live = unit < n_units;
    const int64_t uc = live ? unit : n_units - 1;
    const int64_t tok = uc / groups; const int head0 = (int)(uc - tok * groups) * HPL;
    const e* dy = reinterpret_cast<const e*>(dy_) + tok * dy_st_t + (int64_t)head0 * dy_st_h + 8 * sub;
    v8 g[HPL], v[HPL];
#pragma unroll
    for (int h = 0; h < HPL; ++h) g[h] = *reinterpret_cast<const v8*>(dy + h * dy_st_h);
    if (w_) {
      const e* x = reinterpret_cast<const e*>(x_) + tok * x_st + (int64_t)head0 * 128 + 8 * sub;
#pragma unroll
      for (int h = 0; h < HPL; ++h) v[h] = DTA_SAVED_LOAD(reinterpret_cast<const v8*>(x + h * 128));
    }
    const float* c = cs + tok * 128 + 8 * (sub & 7);
    float cj[8], sj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { cj[j] = c[j]; sj[j] = sub < 8 ? c[64 + j] : -c[64 + j]; }
    e* dx = reinterpret_cast<e*>(dx_) + tok * dx_st + (int64_t)head0 * 128 + 8 * sub;
#pragma unroll
    for (int h = 0; h < HPL; ++h) {
      float da[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gj = (float)g[h][j];
        const float other = __shfl_xor(gj, 8);
        da[j] = gj * cj[j] + other * sj[j];
      }
      v8 o;
      if (w_) {
        const float r = rstd[tok * NH + head0 + h];
        float dot = 0.f, t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { t[j] = (float)v[h][j] * r; dot = __builtin_fmaf(da[j] * (float)wv[j], t[j], dot); if (live) acc[j] = __builtin_fmaf(da[j], t[j], acc[j]); }
#pragma unroll
        for (int o2 = 8; o2 > 0; o2 >>= 1) dot += __shfl_xor(dot, o2);
        dot *= (1.f / 128.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (e)(r * (da[j] * (float)wv[j] - t[j] * dot));
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (e)da[j];
      }
      if (live) *reinterpret_cast<v8*>(dx + h * 128) = o;
    }
  }
  if (w_) {                                              // dw partial [gridDim.x, 128]
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 128) {
      const int s = threadIdx.x >> 3, j = threadIdx.x & 7;                   // element 8*s + j
      float t = 0.f;
      for (int g2 = 0; g2 < 16; ++g2) t += red[(g2 * 16 + s) * 8 + j];
      dw_part[(int64_t)blockIdx.x * 128 + threadIdx.x] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// SwiGLU on rows of C columns; gate/up (and their gradients) may live side by side in one fused
// [rows, 2C] GEMM output: `ld` = elements between consecutive rows of g/u (dg/du).
template <int DT>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const void* __restrict__ g_, const void* __restrict__ u_, void* __restrict__ y_,
                                                         int64_t n8, int c8, int64_t ld) {
  using e = typename ETy<DT>::e; using v8 = typename ETy<DT>::v8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / c8; const int col = (int)(i - row * c8) * 8;
    const v8 g = *reinterpret_cast<const v8*>(reinterpret_cast<const e*>(g_) + row * ld + col);
    const v8 u = *reinterpret_cast<const v8*>(reinterpret_cast<const e*>(u_) + row * ld + col);
    v8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float x = (float)g[j]; const e s = (e)(x / (1.f + __expf(-x))); o[j] = (e)((float)s * (float)u[j]); }
    reinterpret_cast<v8*>(y_)[i] = o;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const void* __restrict__ g_, const void* __restrict__ u_, const void* __restrict__ dy_,
                                                         void* __restrict__ dg_, void* __restrict__ du_, int64_t n8, int c8, int64_t ld, int64_t ldg) {
  using e = typename ETy<DT>::e; using v8 = typename ETy<DT>::v8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / c8; const int col = (int)(i - row * c8) * 8;
    const v8 g = DTA_SAVED_LOAD(reinterpret_cast<const v8*>(reinterpret_cast<const e*>(g_) + row * ld + col));
    const v8 u = DTA_SAVED_LOAD(reinterpret_cast<const v8*>(reinterpret_cast<const e*>(u_) + row * ld + col));
    const v8 dy = reinterpret_cast<const v8*>(dy_)[i];
    v8 dg, du;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (float)g[j], sg = 1.f / (1.f + __expf(-x)), d = (float)dy[j];
      du[j] = (e)(d * x * sg);
      dg[j] = (e)(d * (float)u[j] * sg * (1.f + x * (1.f - sg)));
    }
    *reinterpret_cast<v8*>(reinterpret_cast<e*>(dg_) + row * ldg + col) = dg;
    *reinterpret_cast<v8*>(reinterpret_cast<e*>(du_) + row * ldg + col) = du;
  }
}

// ---------------------------------------------------------------------------------------------
// 2-D transpose through a 64x64 LDS tile of 32-bit words, 16-byte global accesses on both sides.
//   4-byte elements: word[c][r] = in[r][c].
//   2-byte elements: a lane loads the 8-element chunks of TWO consecutive rows and packs (in[r][c], in[r+1][c]) into one word, so the tile
//   is transposed at word granularity ([c][r/2], 32 words per output row) and the write phase reads 4 consecutive words = 8 output
//   elements with one ds_read_b128.  Row pitch = odd number of words (+1 word of padding x alignment): the column-wise writes of the
//   load phase spread over the banks.
template <int ESZ>
__global__ __launch_bounds__(256) void transpose_kernel(const char* __restrict__ in, char* __restrict__ out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out) {
  constexpr int TS = 64;
  constexpr int WPR = ESZ == 2 ? TS / 2 : TS;         // words per transposed tile row
  constexpr int PITCH = WPR + 4;                      // words; +4 keeps 16-byte alignment of the rows for the b128 reads (2-way conflicts at worst)
  __shared__ __attribute__((aligned(16))) uint32_t tile[TS * PITCH];
  const int64_t r0 = (int64_t)blockIdx.y * TS, c0 = (int64_t)blockIdx.x * TS;
  const int tid = threadIdx.x;
  if (ESZ == 2) {
    // load: 32 row pairs x 8 chunks of 8 columns = 256 items, one per thread
    const int rp = tid >> 3, ch = tid & 7, r = 2 * rp, c = 8 * ch;
    uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
    if (c0 + c < cols) {
      if (r0 + r < rows) a = *reinterpret_cast<const uint4*>(in + ((r0 + r) * ld_in + c0 + c) * 2);
      if (r0 + r + 1 < rows) b = *reinterpret_cast<const uint4*>(in + ((r0 + r + 1) * ld_in + c0 + c) * 2);
    }
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {                     // columns c+2j (low halves) and c+2j+1 (high halves)
      tile[(c + 2 * j) * PITCH + rp] = (aw[j] & 0xffffu) | (bw[j] << 16);
      tile[(c + 2 * j + 1) * PITCH + rp] = (aw[j] >> 16) | (bw[j] & 0xffff0000u);
    }
    __syncthreads();
    // store: 64 output rows (input columns) x 8 chunks of 8 output elements (4 words)
    for (int i = tid; i < TS * 8; i += 256) {
      const int oc = i >> 3, k = i & 7;               // output row oc (= input column), output elements 8k .. 8k+7 (= input rows)
      if (c0 + oc < cols && r0 + 8 * k < rows)
        *reinterpret_cast<uint4*>(out + ((c0 + oc) * ld_out + r0 + 8 * k) * 2) = *reinterpret_cast<const uint4*>(&tile[oc * PITCH + 4 * k]);
    }
  } else {
    for (int i = tid; i < TS * 16; i += 256) {        // 64 rows x 16 chunks of 4 columns
      const int r = i >> 4, c = 4 * (i & 15);
      uint4 a = make_uint4(0, 0, 0, 0);
      if (r0 + r < rows && c0 + c < cols) a = *reinterpret_cast<const uint4*>(in + ((r0 + r) * ld_in + c0 + c) * 4);
      tile[(c + 0) * PITCH + r] = a.x; tile[(c + 1) * PITCH + r] = a.y; tile[(c + 2) * PITCH + r] = a.z; tile[(c + 3) * PITCH + r] = a.w;
    }
    __syncthreads();
    for (int i = tid; i < TS * 16; i += 256) {
      const int oc = i >> 4, k = i & 15;
      if (c0 + oc < cols && r0 + 4 * k < rows)
        *reinterpret_cast<uint4*>(out + ((c0 + oc) * ld_out + r0 + 4 * k) * 4) = *reinterpret_cast<const uint4*>(&tile[oc * PITCH + 4 * k]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// out[i] = cast( sum_s part[s * stride + i]  (+ extra[i]) ),  fp32 sums: the reduction of the per-workgroup weight-gradient partials of
// the norm kernels above (many slabs of H values) and of the split-K weight-gradient GEMM (a few slabs of out*in values), fused with the
// rounding to the parameter dtype.
//   tall form: a workgroup of 16 waves owns 64 columns; wave w sums slabs w, w+16, ... (8 loads in flight), LDS combines the 16.
template <int DT>
__global__ __launch_bounds__(1024) void sum_slabs_tall_kernel(const float* __restrict__ part, int64_t slabs, int64_t n, int64_t stride,
                                                              const float* __restrict__ extra, void* __restrict__ out_) {
  using e = typename ETy<DT>::e;
  __shared__ float red[16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + lane;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  if (col < n) {
    int64_t s = wave;
    for (; s + 16 * 7 < slabs; s += 16 * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += part[(s + 16 * u) * stride + col];
    }
    for (; s < slabs; s += 16) acc[0] += part[s * stride + col];
  }
  red[wave * 64 + lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (wave == 0 && col < n) {
    float t = extra ? extra[col] : 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w * 64 + lane];
    reinterpret_cast<e*>(out_)[col] = (e)t;
  }
}
//   flat form (slabs <= 16): one output element group of 4 per lane, slabs summed in order.
template <int DT>
__global__ __launch_bounds__(256) void sum_slabs_flat_kernel(const float* __restrict__ part, int slabs, int64_t n4, int64_t stride,
                                                             const float* __restrict__ extra, void* __restrict__ out_) {
  using e = typename ETy<DT>::e;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef e e4 __attribute__((ext_vector_type(4)));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f4 t = extra ? *reinterpret_cast<const f4*>(extra + 4 * i) : f4{0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; s2 < slabs; ++s2) t += *reinterpret_cast<const f4*>(part + s2 * stride + 4 * i);
    e4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (e)t[j];
    *reinterpret_cast<e4*>(reinterpret_cast<e*>(out_) + 4 * i) = o;
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int row_blocks(int64_t rows, int per_block, int cap) { int64_t b = (rows + per_block - 1) / per_block; return (int)(b < cap ? (b > 0 ? b : 1) : cap); }

}  // namespace

#define DTA_DISPATCH(KERNEL, GRID, ...)                                                                    \
  do { hipStream_t st_ = static_cast<hipStream_t>(stream); DTA_REFUSE_IF_PRIOR_ERROR();                    \
       if (dtype == DTA_BF16) hipLaunchKernelGGL(KERNEL<DTA_BF16>, dim3(GRID), dim3(256), 0, st_, __VA_ARGS__); \
       else if (dtype == DTA_F16) hipLaunchKernelGGL(KERNEL<DTA_F16>, dim3(GRID), dim3(256), 0, st_, __VA_ARGS__); \
       else hipLaunchKernelGGL(KERNEL<DTA_F32>, dim3(GRID), dim3(256), 0, st_, __VA_ARGS__);               \
       return DTA_LAUNCH_STATUS(); } while (0)

extern "C" int dta_rmsnorm_fwd(const void* x, const void* delta, const void* w, void* x_out, void* y, float* rstd,
                               int32_t R, int32_t H, float eps, int32_t dtype, void* stream) {
  if (!x || !w || !y || !rstd || R <= 0 || H <= 0 || ((delta != nullptr) != (x_out != nullptr))) return DTA_EINVAL;
  if (!row_dtype_ok(dtype) || H % 8) return DTA_EUNSUPPORTED;
  if (!al16(x) || !al16(w) || !al16(y) || (delta && (!al16(delta) || !al16(x_out)))) return DTA_EALIGN;
  hipStream_t st_ = static_cast<hipStream_t>(stream);
  DTA_REFUSE_IF_PRIOR_ERROR();
  const dim3 grid(row_blocks(R, 4, 8192)), block(256);
#define DTA_RMS_FWD(NA_)                                                                                                       \
  do { if (dtype == DTA_BF16) hipLaunchKernelGGL((rmsnorm_fwd_kernel<DTA_BF16, NA_>), grid, block, 0, st_, x, delta, w, x_out, y, rstd, R, H, eps); \
       else if (dtype == DTA_F16) hipLaunchKernelGGL((rmsnorm_fwd_kernel<DTA_F16, NA_>), grid, block, 0, st_, x, delta, w, x_out, y, rstd, R, H, eps); \
       else hipLaunchKernelGGL((rmsnorm_fwd_kernel<DTA_F32, NA_>), grid, block, 0, st_, x, delta, w, x_out, y, rstd, R, H, eps); } while (0)
  if (H <= 1024) DTA_RMS_FWD(2);                 // rows of up to 1 024 / 2 048 / 4 096 elements stay in registers between the two passes
  else if (H <= 2048) DTA_RMS_FWD(4);
  else if (H <= 4096) DTA_RMS_FWD(8);
  else DTA_RMS_FWD(0);
#undef DTA_RMS_FWD
  return DTA_LAUNCH_STATUS();
}

/* dw_partial: float [dta_rmsnorm_bwd_blocks(R), H]; the caller sums it over dim 0. */
extern "C" int dta_rmsnorm_bwd_blocks(int32_t R) { return row_blocks(R, 4, 2048); }
extern "C" int dta_rmsnorm_bwd(const void* x, const void* w, const void* dy, const void* dres, const float* rstd, void* dx, float* dw_partial,
                               int32_t R, int32_t H, int32_t dtype, void* stream) {
  if (!x || !w || !dy || !rstd || !dx || !dw_partial || R <= 0 || H <= 0) return DTA_EINVAL;
  if (!row_dtype_ok(dtype) || H % 8 || H > 8192) return DTA_EUNSUPPORTED;
  if (!al16(x) || !al16(w) || !al16(dy) || !al16(dx) || (dres && !al16(dres))) return DTA_EALIGN;
  hipStream_t st_ = static_cast<hipStream_t>(stream);
  DTA_REFUSE_IF_PRIOR_ERROR();
  const dim3 grid(row_blocks(R, 4, 2048)), block(256);
#define DTA_RMS_BWD(NA_)                                                                                                       \
  do { if (dtype == DTA_BF16) hipLaunchKernelGGL((rmsnorm_bwd_kernel<DTA_BF16, NA_>), grid, block, 0, st_, x, w, dy, dres, rstd, dx, dw_partial, R, H); \
       else if (dtype == DTA_F16) hipLaunchKernelGGL((rmsnorm_bwd_kernel<DTA_F16, NA_>), grid, block, 0, st_, x, w, dy, dres, rstd, dx, dw_partial, R, H); \
       else hipLaunchKernelGGL((rmsnorm_bwd_kernel<DTA_F32, NA_>), grid, block, 0, st_, x, w, dy, dres, rstd, dx, dw_partial, R, H); } while (0)
  if (H <= 1024) DTA_RMS_BWD(2);                 // NA = 16-byte groups per lane; the row stays in registers where that fits (see KEEP)
  else if (H <= 2048) DTA_RMS_BWD(4);
  else if (H <= 4096) DTA_RMS_BWD(8);
  else DTA_RMS_BWD(16);
#undef DTA_RMS_BWD
  return DTA_LAUNCH_STATUS();
}

extern "C" int dta_qk_norm_rope_fwd(const void* x, const void* w, const float* cos_sin, void* y, float* rstd,
                                    int32_t T, int32_t NH, int32_t head_dim, int64_t x_stride_t, float eps, int32_t dtype, void* stream) {
  if (!x || !cos_sin || !y || T <= 0 || NH <= 0 || (w && !rstd)) return DTA_EINVAL;
  if (!row_dtype_ok(dtype) || head_dim != 128) return DTA_EUNSUPPORTED;
  if (!al16(x) || !al16(y) || (w && !al16(w)) || x_stride_t % 8) return DTA_EALIGN;
  hipStream_t st_ = static_cast<hipStream_t>(stream);
  DTA_REFUSE_IF_PRIOR_ERROR();
#define DTA_QK_FWD(HPL_)                                                                                                       \
  do { const int64_t n = (int64_t)T * (NH / HPL_); const dim3 grid((unsigned)((n + 15) / 16)), block(256);                   \
       if (dtype == DTA_BF16) hipLaunchKernelGGL((qk_norm_rope_fwd_kernel<DTA_BF16, HPL_>), grid, block, 0, st_, x, w, cos_sin, y, rstd, n, NH, x_stride_t, eps); \
       else if (dtype == DTA_F16) hipLaunchKernelGGL((qk_norm_rope_fwd_kernel<DTA_F16, HPL_>), grid, block, 0, st_, x, w, cos_sin, y, rstd, n, NH, x_stride_t, eps); \
       else hipLaunchKernelGGL((qk_norm_rope_fwd_kernel<DTA_F32, HPL_>), grid, block, 0, st_, x, w, cos_sin, y, rstd, n, NH, x_stride_t, eps); } while (0)
  if (NH % 4 == 0) DTA_QK_FWD(4); else DTA_QK_FWD(1);
#undef DTA_QK_FWD
  return DTA_LAUNCH_STATUS();
}

/* dw_partial: float [dta_qk_norm_rope_bwd_blocks(T*NH), 128] (ignored when w == NULL). */
extern "C" int dta_qk_norm_rope_bwd_blocks(int64_t n_heads_total) { return row_blocks(n_heads_total, 16, 1024); }
extern "C" int dta_qk_norm_rope_bwd(const void* x, const void* w, const float* cos_sin, const void* dy, const float* rstd,
                                    void* dx, float* dw_partial, int32_t T, int32_t NH, int32_t head_dim,
                                    int64_t x_stride_t, int64_t dy_stride_t, int64_t dy_stride_h, int64_t dx_stride_t, int32_t dtype, void* stream) {
  if (!cos_sin || !dy || !dx || T <= 0 || NH <= 0 || (w && (!x || !rstd || !dw_partial))) return DTA_EINVAL;
  if (!row_dtype_ok(dtype) || head_dim != 128) return DTA_EUNSUPPORTED;
  if (!al16(dy) || !al16(dx) || (w && (!al16(w) || !al16(x))) || x_stride_t % 8 || dy_stride_t % 8 || dy_stride_h % 8 || dx_stride_t % 8) return DTA_EALIGN;
  if (dx_stride_t < (int64_t)NH * 128) return DTA_EINVAL;
  hipStream_t st_ = static_cast<hipStream_t>(stream);
  DTA_REFUSE_IF_PRIOR_ERROR();
  // the grid - and with it the number of dw_partial rows the caller sized from dta_qk_norm_rope_bwd_blocks(T*NH) - does not depend on HPL
  const dim3 grid(row_blocks((int64_t)T * NH, 16, 1024)), block(256);
#define DTA_QK_BWD(HPL_)                                                                                                       \
  do { const int64_t n = (int64_t)T * (NH / HPL_);                                                                            \
       if (dtype == DTA_BF16) hipLaunchKernelGGL((qk_norm_rope_bwd_kernel<DTA_BF16, HPL_>), grid, block, 0, st_, x, w, cos_sin, dy, rstd, dx, dw_partial, n, NH, x_stride_t, dy_stride_t, dy_stride_h, dx_stride_t); \
       else if (dtype == DTA_F16) hipLaunchKernelGGL((qk_norm_rope_bwd_kernel<DTA_F16, HPL_>), grid, block, 0, st_, x, w, cos_sin, dy, rstd, dx, dw_partial, n, NH, x_stride_t, dy_stride_t, dy_stride_h, dx_stride_t); \
       else hipLaunchKernelGGL((qk_norm_rope_bwd_kernel<DTA_F32, HPL_>), grid, block, 0, st_, x, w, cos_sin, dy, rstd, dx, dw_partial, n, NH, x_stride_t, dy_stride_t, dy_stride_h, dx_stride_t); } while (0)
  if (NH % 4 == 0) DTA_QK_BWD(4); else DTA_QK_BWD(1);
#undef DTA_QK_BWD
  return DTA_LAUNCH_STATUS();
}

extern "C" int dta_swiglu_fwd(const void* gate, const void* up, void* y, int64_t rows, int32_t cols, int64_t ld, int32_t dtype, void* stream) {
  if (!gate || !up || !y || rows <= 0 || cols <= 0 || ld < cols) return DTA_EINVAL;
  if (!row_dtype_ok(dtype) || cols % 8 || ld % 8) return DTA_EUNSUPPORTED;
  if (!al16(gate) || !al16(up) || !al16(y)) return DTA_EALIGN;
  const int64_t n8 = rows * (cols / 8);
  DTA_DISPATCH(swiglu_fwd_kernel, row_blocks(n8, 256, 4096), gate, up, y, n8, cols / 8, ld);
}

extern "C" int dta_swiglu_bwd(const void* gate, const void* up, const void* dy, void* dgate, void* dup,
                              int64_t rows, int32_t cols, int64_t ld, int64_t ld_grad, int32_t dtype, void* stream) {
  if (!gate || !up || !dy || !dgate || !dup || rows <= 0 || cols <= 0 || ld < cols || ld_grad < cols) return DTA_EINVAL;
  if (!row_dtype_ok(dtype) || cols % 8 || ld % 8 || ld_grad % 8) return DTA_EUNSUPPORTED;
  if (!al16(gate) || !al16(up) || !al16(dy) || !al16(dgate) || !al16(dup)) return DTA_EALIGN;
  const int64_t n8 = rows * (cols / 8);
  DTA_DISPATCH(swiglu_bwd_kernel, row_blocks(n8, 256, 4096), gate, up, dy, dgate, dup, n8, cols / 8, ld, ld_grad);
}


extern "C" int dta_transpose(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, int32_t elem_size, void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0 || ld_in < cols || ld_out < rows) return DTA_EINVAL;
  if (elem_size != 2 && elem_size != 4) return DTA_EUNSUPPORTED;
  const int V = 16 / elem_size;
  if (rows % V || cols % V || ld_in % V || ld_out % V) return DTA_EUNSUPPORTED;
  if (!al16(in) || !al16(out)) return DTA_EALIGN;
  hipStream_t st_ = static_cast<hipStream_t>(stream);
  DTA_REFUSE_IF_PRIOR_ERROR();
  const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64)), block(256);
  if (grid.y > 65535) return DTA_EUNSUPPORTED;
  if (elem_size == 2) hipLaunchKernelGGL(transpose_kernel<2>, grid, block, 0, st_, (const char*)in, (char*)out, rows, cols, ld_in, ld_out);
  else hipLaunchKernelGGL(transpose_kernel<4>, grid, block, 0, st_, (const char*)in, (char*)out, rows, cols, ld_in, ld_out);
  return DTA_LAUNCH_STATUS();
}

extern "C" int dta_sum_slabs(const float* part, int64_t slabs, int64_t n, int64_t slab_stride, const float* extra, void* out, int32_t out_dtype,
                             void* stream) {
  if (!part || !out || slabs <= 0 || n <= 0 || slab_stride < n) return DTA_EINVAL;
  if (!row_dtype_ok(out_dtype)) return DTA_EUNSUPPORTED;
  hipStream_t st_ = static_cast<hipStream_t>(stream);
  DTA_REFUSE_IF_PRIOR_ERROR();
  const bool flat = slabs <= 16 && n % 4 == 0 && slab_stride % 4 == 0 && al16(part) && al16(out) && (!extra || al16(extra));
  if (flat) {
    const dim3 grid(row_blocks(n / 4, 256, 8192)), block(256);
    if (out_dtype == DTA_BF16) hipLaunchKernelGGL(sum_slabs_flat_kernel<DTA_BF16>, grid, block, 0, st_, part, (int)slabs, n / 4, slab_stride, extra, out);
    else if (out_dtype == DTA_F16) hipLaunchKernelGGL(sum_slabs_flat_kernel<DTA_F16>, grid, block, 0, st_, part, (int)slabs, n / 4, slab_stride, extra, out);
    else hipLaunchKernelGGL(sum_slabs_flat_kernel<DTA_F32>, grid, block, 0, st_, part, (int)slabs, n / 4, slab_stride, extra, out);
  } else {
    if ((n + 63) / 64 > 0x7fffffff) return DTA_EUNSUPPORTED;
    const dim3 grid((unsigned)((n + 63) / 64)), block(1024);
    if (out_dtype == DTA_BF16) hipLaunchKernelGGL(sum_slabs_tall_kernel<DTA_BF16>, grid, block, 0, st_, part, slabs, n, slab_stride, extra, out);
    else if (out_dtype == DTA_F16) hipLaunchKernelGGL(sum_slabs_tall_kernel<DTA_F16>, grid, block, 0, st_, part, slabs, n, slab_stride, extra, out);
    else hipLaunchKernelGGL(sum_slabs_tall_kernel<DTA_F32>, grid, block, 0, st_, part, slabs, n, slab_stride, extra, out);
  }
  return DTA_LAUNCH_STATUS();
}
