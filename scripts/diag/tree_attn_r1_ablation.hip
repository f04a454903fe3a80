// DIAGNOSTIC BUILD ONLY (not part of libdta_mi355x.so): the round-1 attention translation unit with its -DDTA_ABL ablation
// switches and the retired 4-wave dK/dV kernel, kept so that the attribution experiments of DESIGN.md §9 can be re-run:
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -DDTA_ABL=<bits> tree_attn_r1_ablation.hip ../trie_kernels.hip ../logprob_kernels.hip ../elementwise_kernels.hip -o /tmp/libdta_abl.so
// and select it with DTA_LIB=/tmp/libdta_abl.so.  Results are wrong for most switch values (timing only).
// Tree attention forward / backward for gfx950 (MI355X, CDNA4).  head_dim = 128, bf16 or f16.
//
// One kernel family serves both forms of the reference's "node attends to its ancestor path":
//   * packed trie (DFS pre-order): key s visible to query t  <=>  s <= t < subtree_end[s]
//   * stack form (tree_training_engine.py:171-186): subtree_end == NULL, q_offset = start
//
// Tiling (wave64, v_mfma_f32_32x32x16):
//   fwd / dQ : workgroup = 4 waves = 128 query rows of one query head; each wave owns 32 rows with the
//              QUERY ON THE MFMA LANE (S^T = K·Q^T), so the softmax row statistics are lane-local and
//              the S^T accumulator is directly the B operand of O^T += V^T·P^T / dQ^T += K^T·dS^T.
//   dK/dV    : workgroup = 4 waves = 128 keys of one kv head; each wave owns 32 keys with the KEY ON
//              THE LANE (S = Q·K^T), dK^T/dV^T live in 128 accumulator registers per wave across the whole
//              query sweep (all query heads of the GQA group) -> no cross-workgroup sum, no atomics.
//   K/V (resp. Q/dO) tiles of 64 rows x 128 cols are staged global -> registers -> LDS (issue early,
//   write late) into one swizzled 256-B-row image that serves BOTH row reads (ds_read_b128) and
//   transposed reads (ds_read_b64_tr_b16).
//
// Lane maps used here were verified on hardware by tests/micro/mfma_layout_probe.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../../include/dta.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native vector (HIP's uint4 class kept staging arrays in scratch)

template <int DT> struct Ty;
template <> struct Ty<DTA_BF16> {
  using e = __bf16; using v8 = bf16x8; using v4 = bf16x4;
  static __device__ __forceinline__ f32x16 mma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Ty<DTA_F16> {
  using e = _Float16; using v8 = f16x8; using v4 = f16x4;
  static __device__ __forceinline__ f32x16 mma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

struct AttnParams {
  const void *q, *k, *v, *o, *dout;
  void *out, *dq, *dk, *dv;
  float *lse_w; const float* lse_r; float* delta;
  const int32_t *subtree_end, *run_ptr, *runs, *ktile_qend;
  const int32_t *dkv_units, *dkv_splits; float* dkv_ws;     // split-Q work units of the dK/dV sweep (NULL: one unit per key tile)
  int32_t Tq, Tk, q_offset, Hq, Hkv, group;
  int64_t q_st, q_sh, kv_st, kv_sh, v_st, v_sh, o_st, o_sh, dq_st, dq_sh, dkv_st, dkv_sh;
  float scale; int32_t accumulate; int32_t ktile;
};

constexpr int TILE_BYTES = 64 * 256;           // 64 rows x 128 x 2 B
constexpr float LOG2E = 1.4426950408889634f;

// Byte offset of 16-B chunk `ch` (0..15) of row `row` in a [rows][128 x 16-bit] image with 256-B rows.
// The XOR makes both the 32x32x16 row reads (ds_read_b128) and the transposed reads conflict-free.
__device__ __forceinline__ int img_off(int row, int ch) {
  return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

template <class V8> __device__ __forceinline__ V8 row_frag(const char* img, int row, int ch) {
  return *reinterpret_cast<const V8*>(img + img_off(row, ch));
}

__device__ __forceinline__ s16x4 tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// A-operand fragment read TRANSPOSED from the image: A[m = 32*mb + (lane&31)][kk], where the 16-deep
// k-step covers image rows R0..R0+15 in the accumulator-as-operand order
// (element j of lane half h <-> image row R0 + 8*(j>>2) + 4*h + (j&3)) and m indexes image columns.
template <class V8> __device__ __forceinline__ V8 tr_frag(const char* img, int R0, int mb, int lane) {
  const int G = lane >> 4, hh = lane >> 5, i = lane & 15, qd = i >> 2, p = i & 3;
  const int ch = 4 * mb + 2 * (G & 1) + (p >> 1);
  const int ra = R0 + 4 * hh + qd;
  s16x4 lo = tr_read(img + img_off(ra, ch) + 8 * (p & 1));
  s16x4 hi = tr_read(img + img_off(ra + 8, ch) + 8 * (p & 1));
  s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(V8, both);
}

// accumulator registers 8*s2 .. 8*s2+7 -> 16-bit fragment of k-step s2 (s2 = 0,1) of a 32-row block
template <int DT> __device__ __forceinline__ typename Ty<DT>::v8 pack_half(const f32x16& x, int s2) {
  typename Ty<DT>::v8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (typename Ty<DT>::e)x[8 * s2 + j];
  return r;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- iterator over the 64-key tiles of a query tile's run list --------------------------------
struct TileIter {
  const int32_t* runs; int ri, re;      // run cursor
  int k0, kend, flag;                   // current tile
  int diag_first_q;                     // NULL-run mode: packed index of the tile's first query
  __device__ __forceinline__ bool load_run() {
    while (ri < re) {
      k0 = __builtin_amdgcn_readfirstlane(runs[4 * ri]); kend = __builtin_amdgcn_readfirstlane(runs[4 * ri + 1]);
      flag = __builtin_amdgcn_readfirstlane(runs[4 * ri + 2]);       // workgroup-uniform: keep the cursor in SGPRs
      if (k0 < kend) return true;
      ++ri;
    }
    return false;
  }
  __device__ __forceinline__ bool advance() {          // to the next tile; false when exhausted
    k0 += 64;
    if (k0 < kend) return true;
    if (runs == nullptr) return false;
    ++ri;
    return load_run();
  }
  __device__ __forceinline__ bool masked() const {
    if (runs == nullptr) return (k0 + 63 > diag_first_q) || (k0 + 64 > kend);
    return flag != 0 || (k0 + 64 > kend);
  }
};

// -------------------------------------------------------------------------------------------------
// Staging macros (no lambdas: captured arrays were demoted to scratch by hipcc).
// A 64-row x 128-col tile pair (A image + B image, 32 KB) is moved by NT threads; every thread owns
// CPT 16-byte chunks of each image: chunk id = tid + NT*i -> row = id >> 4, chunk-in-row = id & 15.
// -------------------------------------------------------------------------------------------------
#define DTA_STAGE_LOAD(REGA, REGB, BASEA, BASEB, STRIDE_A, STRIDE_B, ROW0, ROWMAX, NT, CPT)                \
  _Pragma("unroll") for (int i_ = 0; i_ < (CPT); ++i_) {                                                  \
    const int id_ = tid + (NT) * i_, row_ = id_ >> 4, ch_ = id_ & 15;                                      \
    int gr_ = (ROW0) + row_; gr_ = gr_ < (ROWMAX) ? gr_ : (ROWMAX) - 1;                                    \
    REGA[i_] = *reinterpret_cast<const u32x4*>((BASEA) + (int64_t)gr_ * (STRIDE_A) + ch_ * 8);             \
    REGB[i_] = *reinterpret_cast<const u32x4*>((BASEB) + (int64_t)gr_ * (STRIDE_B) + ch_ * 8);             \
  }
#define DTA_STAGE_WRITE(REGA, REGB, IMGA, IMGB, NT, CPT)                                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < (CPT); ++i_) {                                                  \
    const int id_ = tid + (NT) * i_, row_ = id_ >> 4, ch_ = id_ & 15;                                      \
    *reinterpret_cast<u32x4*>((IMGA) + img_off(row_, ch_)) = REGA[i_];                                     \
    *reinterpret_cast<u32x4*>((IMGB) + img_off(row_, ch_)) = REGB[i_];                                     \
  }

constexpr int SE_BYTES = 256;                                       // 64 x int32 subtree_end of the staged keys
constexpr int QK_LDS = 2 * (2 * TILE_BYTES + SE_BYTES);            // double-buffered {K image, V image, se}

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// Diagnostic ablation switches (-DDTA_ABL=bits; product builds leave them off): 1 = no exp, 2 = no K row reads,
// 4 = no V transposed reads.  They only exist to attribute time (cdna guide, rule 17); results are wrong when set.
#ifndef DTA_ABL
#define DTA_ABL 0
#endif
#define DTA_ABL_E(real, fake) ((DTA_ABL & 1) ? (fake) : (real))
#define DTA_ABL_A(real, fake) ((DTA_ABL & 2) ? (fake) : (real))
#define DTA_ABL_B(real, fake) ((DTA_ABL & 4) ? (fake) : (real))
// 8 = no per-tile barrier in the forward (races: timing only); 16 = forward with one head per workgroup (valid results); 128 = the 4-wave dK/dV kernel instead of the 8-wave one (valid results); 256 / 512 = 8-wave dK/dV without its per-tile barrier / without the K,V fragment reads (timing only)

// Per-lane byte offsets of every fragment read inside one image, computed once: the XOR swizzle depends on the
// lane only (row blocks of 32 and k-steps of 16 rows leave row&3 and (row>>2)&3 unchanged), so inside the tile
// loop every ds_read is <lane offset register> + <compile-time immediate>.
struct FragOffs { int row[8]; int tr[8]; };
__device__ __forceinline__ FragOffs frag_offsets(int lane) {
  FragOffs o;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s = 0; s < 8; ++s) o.row[s] = img_off(r, 2 * s + h);
  const int G = lane >> 4, i = lane & 15, qd = i >> 2, pp = i & 3;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int ch = 4 * mb + 2 * (G & 1) + (pp >> 1);
    o.tr[mb] = img_off(4 * h + qd, ch) + 8 * (pp & 1);
    o.tr[4 + mb] = img_off(4 * h + qd + 8, ch) + 8 * (pp & 1);
  }
  return o;
}
template <class V8> __device__ __forceinline__ V8 tr_pair(const char* lo_p, const char* hi_p) {
  s16x4 lo = tr_read(lo_p);
  s16x4 hi = tr_read(hi_p);
  s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(V8, both);
}
template <class V8> __device__ __forceinline__ V8 tr_frag_o(const char* img_r0, const FragOffs& o, int mb) {
  s16x4 lo = tr_read(img_r0 + o.tr[mb]);
  s16x4 hi = tr_read(img_r0 + o.tr[4 + mb]);
  s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(V8, both);
}

// K/V tile pair global -> LDS by LDS-DMA: NW waves move the 16 + 16 one-KiB pieces (4 image rows each) of the
// two 64-row images; the image's XOR swizzle goes on the per-lane SOURCE chunk.  Per-lane source offsets
// (voff_k / voff_v, bytes inside a 64-row tile) are fixed for the whole sweep, so per tile only a scalar base
// moves (scalar-base + lane-offset DMA form).  subtree_end of the 64 keys goes by 4-byte DMA from wave 0 (an
// ordinary load + ds_write would park that wave for a memory latency on every tile); keys at or beyond the
// run end are excluded by the caller's `k <= min(q, kend-1)` test, not by a sentinel.
#define DTA_KV_OFFSETS(NW)                                                                                 \
  uint32_t voff_k[16 / (NW)], voff_v[16 / (NW)];                                                           \
  _Pragma("unroll") for (int i_ = 0; i_ < 16 / (NW); ++i_) {                                               \
    const int piece_ = wave * (16 / (NW)) + i_, row_ = 4 * piece_ + (lane >> 4);                           \
    const int ch_ = (lane & 15) ^ (((row_ & 3) << 2) | ((row_ >> 2) & 3));                                  \
    voff_k[i_] = (uint32_t)((row_ * p.kv_st + ch_ * 8) * (int64_t)sizeof(e));                              \
    voff_v[i_] = (uint32_t)((row_ * p.v_st + ch_ * 8) * (int64_t)sizeof(e)); }
#define DTA_KV_DMA(BASE, K0, NW)                                                                           \
  { char* base_ = (BASE); const int k0_ = (K0);                                                            \
    if (wave == 0) {                                                                                       \
      if (p.subtree_end) { int ki_ = k0_ + lane; ki_ = ki_ < p.Tk ? ki_ : p.Tk - 1;                        \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.subtree_end + ki_), \
                                         (__attribute__((address_space(3))) void*)(base_ + 2 * TILE_BYTES), 4, 0, 0); } \
      else reinterpret_cast<int*>(base_ + 2 * TILE_BYTES)[lane] = 0x7fffffff; }                            \
    const char* kb_ = reinterpret_cast<const char*>(kbase) + (int64_t)k0_ * p.kv_st * (int64_t)sizeof(e);  \
    const char* vb_ = reinterpret_cast<const char*>(vbase) + (int64_t)k0_ * p.v_st * (int64_t)sizeof(e);   \
    if (k0_ + 64 <= p.Tk) {                                                                                \
      _Pragma("unroll") for (int i_ = 0; i_ < 16 / (NW); ++i_) {                                           \
        const int piece_ = wave * (16 / (NW)) + i_;                                                        \
        uint32_t ok_ = voff_k[i_], ov_ = voff_v[i_];                                                       \
        asm volatile("" : "+v"(ok_), "+v"(ov_));   /* keeps the 32->64-bit extension next to the DMA: scalar-base form */ \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb_ + ok_),       \
                                         (__attribute__((address_space(3))) void*)(base_ + piece_ * 1024), 16, 0, 0);               \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb_ + ov_),       \
                                         (__attribute__((address_space(3))) void*)(base_ + TILE_BYTES + piece_ * 1024), 16, 0, 0); } \
    } else {                                       /* ragged last tile of the tensor: clamp the row per lane */ \
      _Pragma("unroll") for (int i_ = 0; i_ < 16 / (NW); ++i_) {                                           \
        const int piece_ = wave * (16 / (NW)) + i_, row_ = 4 * piece_ + (lane >> 4);                       \
        const int ch_ = (lane & 15) ^ (((row_ & 3) << 2) | ((row_ >> 2) & 3));                              \
        const int rr_ = k0_ + row_ < p.Tk ? row_ : p.Tk - 1 - k0_;                                         \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb_ + ((int64_t)rr_ * p.kv_st + ch_ * 8) * (int64_t)sizeof(e)), \
                                         (__attribute__((address_space(3))) void*)(base_ + piece_ * 1024), 16, 0, 0);               \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb_ + ((int64_t)rr_ * p.v_st + ch_ * 8) * (int64_t)sizeof(e)),  \
                                         (__attribute__((address_space(3))) void*)(base_ + TILE_BYTES + piece_ * 1024), 16, 0, 0); } } }

// =================================================================================================
// forward.  HPB = query heads of one kv group handled by a workgroup (waves 4*hb .. 4*hb+3 own head hb);
// they share the staged K/V tiles.  One barrier per 64-key tile, LDS double buffered, tile loop unrolled
// over the two buffers so that every LDS address is lane-offset + immediate.
// =================================================================================================
template <int DT, int HPB>
__global__ __launch_bounds__(256 * HPB, 2) void tree_attn_fwd_kernel(AttnParams p) {
  using T = Ty<DT>; using e = typename T::e; using v8 = typename T::v8; using v4 = typename T::v4;
  constexpr int NW = 4 * HPB, BUF = 2 * TILE_BYTES + SE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[QK_LDS];

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform values live in SGPRs
  const int hb = wave >> 2, rw = wave & 3;
  const int bid = blockIdx.x;
  const int hgroups = p.group / HPB;
  const int kvh = bid % p.Hkv; const int rest = bid / p.Hkv; const int hgb = rest % hgroups;
  const int nqt = (p.Tq + DTA_QTILE - 1) / DTA_QTILE;
  const int qt = nqt - 1 - rest / hgroups;                          // deepest (heaviest) query tiles first
  const int hq = kvh * p.group + hgb * HPB + hb;
  const int q0 = qt * DTA_QTILE;
  const int qrow = q0 + rw * 32 + r;
  const int qrow_c = qrow < p.Tq ? qrow : p.Tq - 1;
  const int qidx = p.q_offset + qrow;

  TileIter it; it.runs = p.runs; it.diag_first_q = p.q_offset + q0;
  if (p.runs) { it.ri = p.run_ptr[qt]; it.re = p.run_ptr[qt + 1]; if (!it.load_run()) return; }
  else { it.ri = 0; it.re = 1; it.k0 = 0; it.flag = 1; int last = p.q_offset + (q0 + DTA_QTILE < p.Tq ? q0 + DTA_QTILE : p.Tq); it.kend = last < p.Tk ? last : p.Tk; if (it.kend <= 0) return; }

  const e* qp = reinterpret_cast<const e*>(p.q) + (int64_t)qrow_c * p.q_st + (int64_t)hq * p.q_sh;
  v8 qf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const v8*>(qp + 16 * s + 8 * h);

  const e* kbase = reinterpret_cast<const e*>(p.k) + (int64_t)kvh * p.kv_sh;
  const e* vbase = reinterpret_cast<const e*>(p.v) + (int64_t)kvh * p.v_sh;
  const FragOffs offs = frag_offsets(lane);
  DTA_KV_OFFSETS(NW)

  f32x16 O[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int g = 0; g < 16; ++g) O[db][g] = 0.f;
  float m = -1e30f, lsum = 0.f;
  const float c = p.scale * LOG2E;

  int ck0 = it.k0, ckend = it.kend; bool cmask = it.masked();
  DTA_KV_DMA(smem, it.k0, NW)
  bool has_next = it.advance();
  __syncthreads();                                                 // hipcc drains the DMA (vmcnt(0)) in front of the barrier

  // one tile out of buffer BUFI (compile-time): prefetch the next tile into the other buffer, S^T, softmax, PV
#define FWD_TILE(BUFI)                                                                                     \
  {                                                                                                        \
    int nk0_ = 0, nkend_ = 0; bool nmask_ = false;                                                         \
    if (has_next) { nk0_ = it.k0; nkend_ = it.kend; nmask_ = it.masked(); DTA_KV_DMA(smem + (1 - (BUFI)) * BUF, it.k0, NW) } \
    const char* Ks = smem + (BUFI) * BUF; const char* Vs = Ks + TILE_BYTES;                                \
    const int* se_s = reinterpret_cast<const int*>(Ks + 2 * TILE_BYTES);                                   \
    f32x16 X[2];                                                                                           \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                     \
      _Pragma("unroll") for (int g = 0; g < 16; ++g) X[kb][g] = 0.f;                                       \
      _Pragma("unroll") for (int s = 0; s < 8; ++s)                                                        \
        X[kb] = T::mma(DTA_ABL_A(*reinterpret_cast<const v8*>(Ks + 8192 * kb + offs.row[s]), qf[(s + 1) & 7]), qf[s], X[kb]); \
    }                                                                                                      \
    if (cmask) {                                                                                           \
      const int qlim = qidx < ckend ? qidx : ckend - 1;      /* keys at or beyond the run end never count */ \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                     \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                 \
          const int kl = 32 * kb + 8 * gq + 4 * h;                                                         \
          const int4 se4 = *reinterpret_cast<const int4*>(se_s + kl);                                      \
          const int sev[4] = {se4.x, se4.y, se4.z, se4.w};                                                 \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                  \
            const bool ok = (ck0 + kl + j <= qlim) && (qidx < sev[j]);                                     \
            X[kb][4 * gq + j] = ok ? X[kb][4 * gq + j] : -INFINITY;                                        \
          }                                                                                                \
        }                                                                                                  \
    }                                                                                                      \
    float mx = max3(X[0][0], X[0][1], X[0][2]);                                                            \
    _Pragma("unroll") for (int g = 3; g < 15; g += 2) mx = max3(mx, X[0][g], X[0][g + 1]);                 \
    mx = fmaxf(mx, X[0][15]);                                                                              \
    _Pragma("unroll") for (int g = 0; g < 16; g += 2) mx = max3(mx, X[1][g], X[1][g + 1]);                 \
    mx = fmaxf(mx, __shfl_xor(mx, 32));                                                                    \
    const float mc = mx * c;                                                                               \
    if (__any(mc > m)) {                       /* O is rescaled only when some row's maximum really grew */ \
      const float mnew = fmaxf(m, mc);                                                                     \
      const float alpha = fast_exp2(m - mnew);                                                             \
      m = mnew; lsum *= alpha;                                                                             \
      _Pragma("unroll") for (int db = 0; db < 4; ++db)                                                     \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) O[db][g] *= alpha;                                  \
    }                                                                                                      \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                       \
      _Pragma("unroll") for (int g = 0; g < 16; ++g) { const float pv = DTA_ABL_E(fast_exp2(__builtin_fmaf(X[kb][g], c, -m)), X[kb][g] * c); lsum += pv; X[kb][g] = pv; } \
    _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                                     \
      const v8 pb = pack_half<DT>(X[s4 >> 1], s4 & 1);                                                     \
      _Pragma("unroll") for (int db = 0; db < 4; ++db) O[db] = T::mma(DTA_ABL_B(tr_frag_o<v8>(Vs + 4096 * s4, offs, db), qf[db + s4]), pb, O[db]); \
    }                                                                                                      \
    if (!(DTA_ABL & 8)) __syncthreads();                                                                   \
    if (!has_next) break;                                                                                  \
    ck0 = nk0_; ckend = nkend_; cmask = nmask_;                                                            \
    has_next = it.advance();                                                                               \
  }
  while (true) {
    FWD_TILE(0)
    FWD_TILE(1)
  }
#undef FWD_TILE

  lsum += __shfl_xor(lsum, 32);
  const float inv = 1.f / lsum;
  if (qrow < p.Tq) {
    e* op = reinterpret_cast<e*>(p.out) + (int64_t)qrow * p.o_st + (int64_t)hq * p.o_sh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        v4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (e)(O[db][4 * gq + j] * inv);
        *reinterpret_cast<v4*>(op + 32 * db + 8 * gq + 4 * h) = w;
      }
    if (h == 0) p.lse_w[(int64_t)hq * p.Tq + qrow] = m + __builtin_amdgcn_logf(lsum);   // v_log_f32 = log2
  }
}

// =================================================================================================
// backward part 1: delta + dQ   (query tile owns the workgroup; same sweep as the forward)
// =================================================================================================
template <int DT, int HPB>
__global__ __launch_bounds__(256 * HPB, HPB == 2 ? 2 : 1) void tree_attn_bwd_dq_kernel(AttnParams p) {
  using T = Ty<DT>; using e = typename T::e; using v8 = typename T::v8; using v4 = typename T::v4;
  constexpr int NT = 256 * HPB, CPT = 1024 / NT;
  __shared__ __attribute__((aligned(16))) char smem[QK_LDS];

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform values live in SGPRs
  const int hb = wave >> 2, rw = wave & 3;
  const int bid = blockIdx.x;
  const int hgroups = p.group / HPB;
  const int kvh = bid % p.Hkv; const int rest = bid / p.Hkv; const int hgb = rest % hgroups;
  const int nqt = (p.Tq + DTA_QTILE - 1) / DTA_QTILE;
  const int qt = nqt - 1 - rest / hgroups;
  const int hq = kvh * p.group + hgb * HPB + hb;
  const int q0 = qt * DTA_QTILE;
  const int qrow = q0 + rw * 32 + r;
  const int qrow_c = qrow < p.Tq ? qrow : p.Tq - 1;
  const int qidx = p.q_offset + qrow;

  const e* qp = reinterpret_cast<const e*>(p.q) + (int64_t)qrow_c * p.q_st + (int64_t)hq * p.q_sh;
  const e* dop = reinterpret_cast<const e*>(p.dout) + (int64_t)qrow_c * p.o_st + (int64_t)hq * p.o_sh;
  const e* op = reinterpret_cast<const e*>(p.o) + (int64_t)qrow_c * p.o_st + (int64_t)hq * p.o_sh;
  v8 qf[8], dof[8];
  float dsum = 0.f;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    qf[s] = *reinterpret_cast<const v8*>(qp + 16 * s + 8 * h);
    dof[s] = *reinterpret_cast<const v8*>(dop + 16 * s + 8 * h);
    const v8 of = *reinterpret_cast<const v8*>(op + 16 * s + 8 * h);
#pragma unroll
    for (int j = 0; j < 8; ++j) dsum += (float)dof[s][j] * (float)of[j];
  }
  dsum += __shfl_xor(dsum, 32);
  const float delta = dsum;
  const float lse2 = p.lse_r[(int64_t)hq * p.Tq + qrow_c];
  if (h == 0 && qrow < p.Tq) p.delta[(int64_t)hq * p.Tq + qrow] = delta;

  TileIter it; it.runs = p.runs; it.diag_first_q = p.q_offset + q0;
  bool any = true;
  if (p.runs) { it.ri = p.run_ptr[qt]; it.re = p.run_ptr[qt + 1]; any = it.load_run(); }
  else { it.ri = 0; it.re = 1; it.k0 = 0; it.flag = 1; int last = p.q_offset + (q0 + DTA_QTILE < p.Tq ? q0 + DTA_QTILE : p.Tq); it.kend = last < p.Tk ? last : p.Tk; any = it.kend > 0; }

  const e* kbase = reinterpret_cast<const e*>(p.k) + (int64_t)kvh * p.kv_sh;
  const e* vbase = reinterpret_cast<const e*>(p.v) + (int64_t)kvh * p.v_sh;
  u32x4 kreg[CPT], vreg[CPT]; int sereg = 0;
  // chunk id = tid + NT*i -> image row id>>4, chunk id&15: the per-lane byte offset inside a 64-row tile is fixed for the
  // sweep, so a tile's loads are <scalar tile base> + <32-bit lane offset> (no per-lane 64-bit address arithmetic)
  uint32_t soff_k[CPT], soff_v[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int id = tid + NT * i, row = id >> 4, ch = id & 15;
    soff_k[i] = (uint32_t)((row * p.kv_st + ch * 8) * (int64_t)sizeof(e));
    soff_v[i] = (uint32_t)((row * p.v_st + ch * 8) * (int64_t)sizeof(e));
  }
#define DQ_LOAD(K0, KEND)                                                                                  \
  { const int k0_ = (K0);                                                                                  \
    if (k0_ + 64 <= p.Tk) {                                                                                \
      const char* kb_ = reinterpret_cast<const char*>(kbase) + (int64_t)k0_ * p.kv_st * (int64_t)sizeof(e); \
      const char* vb_ = reinterpret_cast<const char*>(vbase) + (int64_t)k0_ * p.v_st * (int64_t)sizeof(e);  \
      _Pragma("unroll") for (int i_ = 0; i_ < CPT; ++i_) {                                                 \
        uint32_t ok_ = soff_k[i_], ov_ = soff_v[i_];                                                       \
        asm volatile("" : "+v"(ok_), "+v"(ov_));   /* keeps the 32->64-bit extension next to the load: scalar-base form */ \
        kreg[i_] = *reinterpret_cast<const u32x4*>(kb_ + ok_);                                             \
        vreg[i_] = *reinterpret_cast<const u32x4*>(vb_ + ov_); }                                           \
    } else { DTA_STAGE_LOAD(kreg, vreg, kbase, vbase, p.kv_st, p.v_st, k0_, p.Tk, NT, CPT) }               \
    if (tid < 64) { const int ki_ = k0_ + tid; sereg = (ki_ < (KEND)) ? (p.subtree_end ? p.subtree_end[ki_] : 0x7fffffff) : 0; } }
#define DQ_WRITE(B)                                                                                        \
  { char* base_ = smem + (B) * (2 * TILE_BYTES + SE_BYTES);                                                \
    DTA_STAGE_WRITE(kreg, vreg, base_, base_ + TILE_BYTES, NT, CPT)                                        \
    if (tid < 64) reinterpret_cast<int*>(base_ + 2 * TILE_BYTES)[tid] = sereg; }

  f32x16 DQ[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int g = 0; g < 16; ++g) DQ[db][g] = 0.f;
  const float c = p.scale * LOG2E;

  const float delta_s = delta * p.scale;
  if (any) {
    int ck0 = it.k0; bool cmask = it.masked();
    DQ_LOAD(it.k0, it.kend) DQ_WRITE(0)
    bool has_next = it.advance();
    int nk0 = it.k0; bool nmask = has_next ? it.masked() : false;
    if (has_next) DQ_LOAD(it.k0, it.kend)
    __syncthreads();
    int cur = 0;
    while (true) {
      bool has_next2 = false;
      if (has_next) {
        DQ_WRITE(cur ^ 1)
        has_next2 = it.advance();
        if (has_next2) DQ_LOAD(it.k0, it.kend)
      }
      const char* Ks = smem + cur * (2 * TILE_BYTES + SE_BYTES); const char* Vs = Ks + TILE_BYTES;
      const int* se_s = reinterpret_cast<const int*>(Ks + 2 * TILE_BYTES);
      // one 32-key block at a time keeps S^T/dP^T at 32 live accumulators (2 waves per SIMD need <= 256 registers)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16 X, DP;
#pragma unroll
        for (int g = 0; g < 16; ++g) { X[g] = 0.f; DP[g] = 0.f; }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          X = T::mma(row_frag<v8>(Ks, 32 * kb + r, 2 * s + h), qf[s], X);
          DP = T::mma(row_frag<v8>(Vs, 32 * kb + r, 2 * s + h), dof[s], DP);
        }
        // dS^T = P ∘ (dP·scale − delta·scale); the interval mask only on tiles of runs flagged partial
        if (cmask) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int kl = 32 * kb + 8 * gq + 4 * h;
            const int4 se4 = *reinterpret_cast<const int4*>(se_s + kl);
            const int sev[4] = {se4.x, se4.y, se4.z, se4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int g = 4 * gq + j;
              const bool ok = (ck0 + kl + j <= qidx) && (qidx < sev[j]);
              const float pv = ok ? fast_exp2(__builtin_fmaf(X[g], c, -lse2)) : 0.f;
              X[g] = pv * __builtin_fmaf(DP[g], p.scale, -delta_s);
            }
          }
        } else {
#pragma unroll
          for (int g = 0; g < 16; ++g) X[g] = fast_exp2(__builtin_fmaf(X[g], c, -lse2)) * __builtin_fmaf(DP[g], p.scale, -delta_s);
        }
        // dQ^T[d][q] += K^T · dS^T
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const v8 db_ = pack_half<DT>(X, s2);
#pragma unroll
          for (int db = 0; db < 4; ++db) DQ[db] = T::mma(tr_frag<v8>(Ks, 32 * kb + 16 * s2, db, lane), db_, DQ[db]);
        }
      }
      __syncthreads();
      if (!has_next) break;
      cur ^= 1; ck0 = nk0; cmask = nmask;
      has_next = has_next2; nk0 = it.k0; nmask = has_next2 ? it.masked() : false;
    }
  }
#undef DQ_LOAD
#undef DQ_WRITE
  if (qrow < p.Tq) {
    e* dqp = reinterpret_cast<e*>(p.dq) + (int64_t)qrow * p.dq_st + (int64_t)hq * p.dq_sh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        v4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (e)DQ[db][4 * gq + j];
        *reinterpret_cast<v4*>(dqp + 32 * db + 8 * gq + 4 * h) = w;
      }
  }
}

// =================================================================================================
// backward part 2: dK, dV.  Key tile owns the workgroup; one wave per SIMD; each wave owns KB blocks of 32
// keys (KB = 2: 64 keys, 256 accumulator registers), so every Q / dO fragment read from LDS (row-wise
// for S, dP and transposed for dK^T, dV^T) feeds KB MFMAs.
// =================================================================================================
constexpr int KV_LDS = 2 * (2 * TILE_BYTES + 512);                 // double-buffered {Q image, dO image, lse[64], delta[64]}

template <int DT, int KB, int NG>
__global__ __launch_bounds__(256 * NG) void tree_attn_bwd_dkv_kernel(AttnParams p) {
  using T = Ty<DT>; using e = typename T::e; using v8 = typename T::v8; using v4 = typename T::v4;
  constexpr int KT = 128 * KB;
  __shared__ __attribute__((aligned(16))) char smem_all[NG * KV_LDS];

  // NG = 2: two groups of 4 waves (two waves per SIMD) own the SAME keys and take alternate items of the
  // (query head, query tile) sweep through their own double-buffered Q/dO images; their partial dK/dV are
  // summed through LDS at the end in a fixed order.
  const int tid = threadIdx.x & 255, grp = NG == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave-uniform values live in SGPRs: scalar address arithmetic
  char* smem = smem_all + grp * KV_LDS;
  const int bid = blockIdx.x;
  const int kvh = bid % p.Hkv; const int unit = bid / p.Hkv;
  // a unit = (key tile, packed query range, slab): heavy key tiles (root-side: every query below them sees
  // them) are cut into several units so that no workgroup carries a serial chain of hundreds of tiles
  const int kt = p.dkv_units ? p.dkv_units[4 * unit] : unit;
  const int slab = p.dkv_units ? p.dkv_units[4 * unit + 3] : -1;
  const int k0 = kt * KT;
  const int q_hi = p.q_offset + p.Tq;
  int kidx[KB], se_l[KB];
  v8 kf[KB][8], vf[KB][8];
#pragma unroll
  for (int b = 0; b < KB; ++b) {
    kidx[b] = k0 + wave * 32 * KB + 32 * b + r;
    const int kc = kidx[b] < p.Tk ? kidx[b] : p.Tk - 1;
    int se = (kidx[b] < p.Tk) ? (p.subtree_end ? p.subtree_end[kidx[b]] : 0x7fffffff) : 0;
    se_l[b] = se < q_hi ? se : q_hi;
    const e* kp = reinterpret_cast<const e*>(p.k) + (int64_t)kc * p.kv_st + (int64_t)kvh * p.kv_sh;
    const e* vp = reinterpret_cast<const e*>(p.v) + (int64_t)kc * p.v_st + (int64_t)kvh * p.v_sh;
#pragma unroll
    for (int s = 0; s < 8; ++s) { kf[b][s] = *reinterpret_cast<const v8*>(kp + 16 * s + 8 * h); vf[b][s] = *reinterpret_cast<const v8*>(vp + 16 * s + 8 * h); }
  }
  // smallest subtree end over the workgroup's keys: query tiles entirely below it (and below the key tile
  // itself) need no mask at all
  __shared__ int se_min_s[4];
  { int mn = se_l[0];
#pragma unroll
    for (int b = 1; b < KB; ++b) mn = se_l[b] < mn ? se_l[b] : mn;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(mn, o); mn = t < mn ? t : mn; }
    if (lane == 0 && grp == 0) se_min_s[wave] = mn; }
  __syncthreads();
  const int se_min = __builtin_amdgcn_readfirstlane(min(min(se_min_s[0], se_min_s[1]), min(se_min_s[2], se_min_s[3])));

  const FragOffs offs = frag_offsets(lane);
  f32x16 DK[KB][4], DV[KB][4];
#pragma unroll
  for (int b = 0; b < KB; ++b)
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 16; ++g) { DK[b][db][g] = 0.f; DV[b][db][g] = 0.f; }

  int qbeg, qend;
  if (p.dkv_units) { qbeg = __builtin_amdgcn_readfirstlane(p.dkv_units[4 * unit + 1]); qend = __builtin_amdgcn_readfirstlane(p.dkv_units[4 * unit + 2]); }
  else {
    qbeg = k0 > p.q_offset ? k0 : p.q_offset;                        // packed index of the first query that can see a key here
    qend = p.ktile_qend ? p.ktile_qend[kt] : q_hi; qend = qend < q_hi ? qend : q_hi;
  }
  const int ntile = qend > qbeg ? (qend - qbeg + 63) / 64 : 0;
  const int total = ntile * p.group;
  const float c = p.scale * LOG2E;

  // Q / dO tiles go global -> LDS directly (LDS-DMA, no staging registers, no ds_write): a wave instruction
  // lands 64 x 16 B = 4 image rows lane-linearly, so the XOR swizzle of the image is applied to the per-lane
  // SOURCE chunk instead (the read side uses the same involution).  lse / delta are 64 floats each and go by
  // 4-byte DMA from waves 0 / 1 (an ordinary load + ds_write would stall those waves for a full memory latency per tile).
  // Per-lane byte offsets of this lane's 16-B chunk inside a 64-row tile are fixed for the whole sweep (piece i of this
  // wave covers image rows 16*wave + 4*i .. +3); per tile only the UNIFORM base moves, so the DMA takes the
  // scalar-base + 32-bit-lane-offset form and costs no per-lane address arithmetic in the loop.
  uint32_t voff_q[4], voff_d[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row_ = 16 * wave + 4 * i + (lane >> 4);
    const int ch_ = (lane & 15) ^ (((lane >> 4) << 2) | i);              // (row_ & 3) = lane >> 4, (row_ >> 2) & 3 = i
    voff_q[i] = (uint32_t)((row_ * p.q_st + ch_ * 8) * (int64_t)sizeof(e));
    voff_d[i] = (uint32_t)((row_ * p.o_st + ch_ * 8) * (int64_t)sizeof(e));
  }
#define KV_DMA(HG, TI, B)                                                                                  \
  { const int hq_ = __builtin_amdgcn_readfirstlane(kvh * p.group + (HG));                                   \
    const int row0_ = qbeg + 64 * (TI) - p.q_offset;                                                       \
    char* base_ = smem + (B) * (2 * TILE_BYTES + 512);                                                     \
    if (wave < 2) { int qr_ = row0_ + lane; qr_ = qr_ < p.Tq ? qr_ : p.Tq - 1;   /* wave 0: lse[64], wave 1: delta[64] */ \
      const float* src_ = (wave == 0 ? p.lse_r : p.delta) + (int64_t)hq_ * p.Tq;                           \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_ + qr_),        \
                                       (__attribute__((address_space(3))) void*)(base_ + 2 * TILE_BYTES + wave * 256), 4, 0, 0); } \
    const char* qb_ = reinterpret_cast<const char*>(p.q) + ((int64_t)hq_ * p.q_sh + (int64_t)row0_ * p.q_st) * (int64_t)sizeof(e);    \
    const char* db_ = reinterpret_cast<const char*>(p.dout) + ((int64_t)hq_ * p.o_sh + (int64_t)row0_ * p.o_st) * (int64_t)sizeof(e); \
    if (row0_ + 64 <= p.Tq) {                                                                              \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                   \
        uint32_t oq_ = voff_q[i_], od_ = voff_d[i_];                                                       \
        asm volatile("" : "+v"(oq_), "+v"(od_));   /* keeps the 32->64-bit extension next to the DMA: scalar-base form */ \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qb_ + oq_),       \
                                         (__attribute__((address_space(3))) void*)(base_ + (wave * 4 + i_) * 1024), 16, 0, 0);              \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db_ + od_),       \
                                         (__attribute__((address_space(3))) void*)(base_ + TILE_BYTES + (wave * 4 + i_) * 1024), 16, 0, 0); } \
    } else {                                       /* ragged last tile of the tensor: clamp the row per lane */ \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                   \
        const int row_ = 16 * wave + 4 * i_ + (lane >> 4);                                                 \
        const int ch_ = (lane & 15) ^ (((lane >> 4) << 2) | i_);                                           \
        const int rr_ = row0_ + row_ < p.Tq ? row_ : p.Tq - 1 - row0_;                                     \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qb_ + ((int64_t)rr_ * p.q_st + ch_ * 8) * (int64_t)sizeof(e)), \
                                         (__attribute__((address_space(3))) void*)(base_ + (wave * 4 + i_) * 1024), 16, 0, 0);              \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db_ + ((int64_t)rr_ * p.o_st + ch_ * 8) * (int64_t)sizeof(e)), \
                                         (__attribute__((address_space(3))) void*)(base_ + TILE_BYTES + (wave * 4 + i_) * 1024), 16, 0, 0); } } }

  const int niter = (total + NG - 1) / NG;               // same barrier count in every group
  {
    int hg_c = 0, ti_c = grp;                        // (query head of the group, query tile) of item idx, advanced without division
    while (ntile > 0 && ti_c >= ntile) { ti_c -= ntile; ++hg_c; }
    if (grp < total) KV_DMA(hg_c, ti_c, 0)
    __syncthreads();                                 // hipcc drains the DMA (vmcnt(0)) in front of the barrier
    int cur = 0;
    for (int it_ = 0; it_ < niter; ++it_) {
      const int idx = it_ * NG + grp;
      const int ti = ti_c;
      ti_c += NG;
      while (ti_c >= ntile) { ti_c -= ntile; ++hg_c; }
      if (idx + NG < total) KV_DMA(hg_c, ti_c, cur ^ 1)  // buffer cur^1 was last read before the previous barrier
      if (NG == 1 || idx < total) {
      const char* Qs = smem + cur * (2 * TILE_BYTES + 512);
      const float* lse_s = reinterpret_cast<const float*>(Qs + 2 * TILE_BYTES); const float* del_s = lse_s + 64;
      const int qi0 = qbeg + 64 * ti;                                   // packed index of image row 0
      const bool full = (qi0 >= k0 + KT - 1) && (qi0 + 63 < se_min);        // workgroup-uniform: no mask needed
      // one 32-row query block at a time (NOT unrolled: both blocks then share one S/dP register set; unrolled, hipcc
      // parks a dK/dV tile in VGPRs around the first block, 64 extra accumulator moves per tile)
#pragma unroll 1
      for (int qb = 0; qb < 2; ++qb) {
        // every LDS read below is <address register> + <immediate>: 16 adds of the (buffer, block) base per block
        const int sb = cur * (2 * TILE_BYTES + 512) + qb * (32 * 256);
        int ar[8], at[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ar[j] = offs.row[j] + sb; at[j] = offs.tr[j] + sb; }
        // S and dP start at 0 (inline constant, no register traffic):  p = exp2(c*S - lse),  dS/scale = p*(dP - delta);
        // the softmax scale of dS is applied once to the dK accumulators in the epilogue
        f32x16 S[KB], DP[KB];
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
          for (int g = 0; g < 16; ++g) { S[b][g] = 0.f; DP[b][g] = 0.f; }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const v8 aq = *reinterpret_cast<const v8*>(smem + ar[s]);
          const v8 ad = *reinterpret_cast<const v8*>(smem + ar[s] + TILE_BYTES);
#pragma unroll
          for (int b = 0; b < KB; ++b) { S[b] = T::mma(aq, kf[b][s], S[b]); DP[b] = T::mma(ad, vf[b][s], DP[b]); }
        }
        // row constants come from LDS only now, after the MFMA chains: they occupy registers for the VALU phase alone
        float nl[16], dl[16];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int ql = 32 * qb + 8 * gq + 4 * h;
          const float4 l4 = *reinterpret_cast<const float4*>(lse_s + ql);
          const float4 d4 = *reinterpret_cast<const float4*>(del_s + ql);
          nl[4 * gq] = l4.x; nl[4 * gq + 1] = l4.y; nl[4 * gq + 2] = l4.z; nl[4 * gq + 3] = l4.w;
          dl[4 * gq] = d4.x; dl[4 * gq + 1] = d4.y; dl[4 * gq + 2] = d4.z; dl[4 * gq + 3] = d4.w;
        }
        if (full) {
#pragma unroll
          for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
              const float pv = fast_exp2(__builtin_fmaf(S[b][g], c, -nl[g]));
              S[b][g] = pv;
              DP[b][g] = pv * (DP[b][g] - dl[g]);
            }
        } else {
#pragma unroll
          for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
              const int qi = qi0 + 32 * qb + 8 * (g >> 2) + 4 * h + (g & 3);
              const bool ok = (kidx[b] <= qi) && (qi < se_l[b]);
              const float pv = ok ? fast_exp2(__builtin_fmaf(S[b][g], c, -nl[g])) : 0.f;
              S[b][g] = pv;
              DP[b][g] = pv * (DP[b][g] - dl[g]);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          v8 pb[KB], sb[KB];
#pragma unroll
          for (int b = 0; b < KB; ++b) { pb[b] = pack_half<DT>(S[b], s2); sb[b] = pack_half<DT>(DP[b], s2); }
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            const v8 adt = tr_pair<v8>(smem + at[db] + TILE_BYTES + 4096 * s2, smem + at[4 + db] + TILE_BYTES + 4096 * s2);
            const v8 aqt = tr_pair<v8>(smem + at[db] + 4096 * s2, smem + at[4 + db] + 4096 * s2);
#pragma unroll
            for (int b = 0; b < KB; ++b) { DV[b][db] = T::mma(adt, pb[b], DV[b][db]); DK[b][db] = T::mma(aqt, sb[b], DK[b][db]); }
          }
        }
      }
      }
      __syncthreads();
      cur ^= 1;
    }
  }
#undef KV_DMA
  if (NG == 2) {
    // group 1 hands its partial sums to group 0 through LDS, 32 accumulators (one d-block of dK and dV) at a time
    float* red = reinterpret_cast<float*>(smem_all);
#pragma unroll
    for (int b = 0; b < KB; ++b)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        if (grp == 1) {
#pragma unroll
          for (int g = 0; g < 16; ++g) { red[g * 256 + tid] = DK[b][db][g]; red[(16 + g) * 256 + tid] = DV[b][db][g]; }
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
          for (int g = 0; g < 16; ++g) { DK[b][db][g] += red[g * 256 + tid]; DV[b][db][g] += red[(16 + g) * 256 + tid]; }
        }
        __syncthreads();
      }
    if (grp == 1) return;
  }
#pragma unroll
  for (int b = 0; b < KB; ++b)
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 16; ++g) DK[b][db][g] *= p.scale;
#pragma unroll
  for (int b = 0; b < KB; ++b) {
    const int kloc = wave * 32 * KB + 32 * b + r;
    if (slab >= 0) {
      // partial sums of a split key tile: fp32 slab [2][KT keys][128 d], summed in unit order by the finalize kernel
      float* ws = p.dkv_ws + ((int64_t)slab * p.Hkv + kvh) * (2 * KT * 128) + (int64_t)kloc * 128;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int d = 32 * db + 8 * gq + 4 * h;
          *reinterpret_cast<float4*>(ws + d) = make_float4(DK[b][db][4 * gq], DK[b][db][4 * gq + 1], DK[b][db][4 * gq + 2], DK[b][db][4 * gq + 3]);
          *reinterpret_cast<float4*>(ws + KT * 128 + d) = make_float4(DV[b][db][4 * gq], DV[b][db][4 * gq + 1], DV[b][db][4 * gq + 2], DV[b][db][4 * gq + 3]);
        }
    } else if (kidx[b] < p.Tk) {
      e* dkp = reinterpret_cast<e*>(p.dk) + (int64_t)kidx[b] * p.dkv_st + (int64_t)kvh * p.dkv_sh;
      e* dvp = reinterpret_cast<e*>(p.dv) + (int64_t)kidx[b] * p.dkv_st + (int64_t)kvh * p.dkv_sh;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int d = 32 * db + 8 * gq + 4 * h;
          v4 wk, wv;
          if (p.accumulate) {
            const v4 ok_ = *reinterpret_cast<const v4*>(dkp + d); const v4 ov_ = *reinterpret_cast<const v4*>(dvp + d);
#pragma unroll
            for (int j = 0; j < 4; ++j) { wk[j] = (e)(DK[b][db][4 * gq + j] + (float)ok_[j]); wv[j] = (e)(DV[b][db][4 * gq + j] + (float)ov_[j]); }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { wk[j] = (e)DK[b][db][4 * gq + j]; wv[j] = (e)DV[b][db][4 * gq + j]; }
          }
          *reinterpret_cast<v4*>(dkp + d) = wk;
          *reinterpret_cast<v4*>(dvp + d) = wv;
        }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// dK/dV with TWO waves per SIMD (8 waves): the two wave groups own the same 128 keys and split every 64-row query tile
// between them (group g takes rows 32g..32g+31), so they share ONE double-buffered Q/dO image.  To fit 256 registers
// the K/V fragments (pure MFMA B operands) live in LDS in fragment order (one lane-linear, conflict-free ds_read_b128
// per use) instead of 64 registers.  The groups' partial dK/dV are summed through LDS in a fixed order at the end.
// -------------------------------------------------------------------------------------------------
constexpr int KV2_FRAGS = 4 * 16384;                                // 4 key slots x {K: 8 fragments x 1 KiB, V: 8 x 1 KiB}
constexpr int KV2_BUF = 2 * TILE_BYTES + 512;
constexpr int KV2_LDS = KV2_FRAGS + 2 * KV2_BUF;

template <int DT>
__global__ __launch_bounds__(512, 2) void tree_attn_bwd_dkv2_kernel(AttnParams p) {
  using T = Ty<DT>; using e = typename T::e; using v8 = typename T::v8; using v4 = typename T::v4;
  constexpr int KT = 128;
  __shared__ __attribute__((aligned(16))) char smem_all[KV2_LDS];
  const int tid8 = threadIdx.x, tid = tid8 & 255, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid8 >> 6);       // 0..7
  const int grp = wave8 >> 2, wave = wave8 & 3;
  char* kvs = smem_all + wave * 16384;                               // this key slot's K fragments (+8192: V)
  char* smem = smem_all + KV2_FRAGS;                                 // the Q/dO buffers
  const int bid = blockIdx.x;
  const int kvh = bid % p.Hkv; const int unit = bid / p.Hkv;
  const int kt = p.dkv_units ? p.dkv_units[4 * unit] : unit;
  const int slab = p.dkv_units ? p.dkv_units[4 * unit + 3] : -1;
  const int k0 = kt * KT;
  const int q_hi = p.q_offset + p.Tq;
  const int kidx = k0 + wave * 32 + r;
  int se_l;
  { const int kc = kidx < p.Tk ? kidx : p.Tk - 1;
    int se = (kidx < p.Tk) ? (p.subtree_end ? p.subtree_end[kidx] : 0x7fffffff) : 0;
    se_l = se < q_hi ? se : q_hi;
    if (grp == 0) {                                                  // group 0 stages the fragments both groups read
      const e* kp = reinterpret_cast<const e*>(p.k) + (int64_t)kc * p.kv_st + (int64_t)kvh * p.kv_sh;
      const e* vp = reinterpret_cast<const e*>(p.v) + (int64_t)kc * p.v_st + (int64_t)kvh * p.v_sh;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        *reinterpret_cast<v8*>(kvs + s * 1024 + lane * 16) = *reinterpret_cast<const v8*>(kp + 16 * s + 8 * h);
        *reinterpret_cast<v8*>(kvs + 8192 + s * 1024 + lane * 16) = *reinterpret_cast<const v8*>(vp + 16 * s + 8 * h);
      }
    } }
  __shared__ int se_min_s[4];
  { int mn = se_l;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(mn, o); mn = t < mn ? t : mn; }
    if (lane == 0 && grp == 0) se_min_s[wave] = mn; }
  __syncthreads();
  const int se_min = __builtin_amdgcn_readfirstlane(min(min(se_min_s[0], se_min_s[1]), min(se_min_s[2], se_min_s[3])));

  const FragOffs offs = frag_offsets(lane);
  f32x16 DK[4], DV[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int g = 0; g < 16; ++g) { DK[db][g] = 0.f; DV[db][g] = 0.f; }

  int qbeg, qend;
  if (p.dkv_units) { qbeg = __builtin_amdgcn_readfirstlane(p.dkv_units[4 * unit + 1]); qend = __builtin_amdgcn_readfirstlane(p.dkv_units[4 * unit + 2]); }
  else {
    qbeg = k0 > p.q_offset ? k0 : p.q_offset;
    qend = p.ktile_qend ? p.ktile_qend[kt] : q_hi; qend = qend < q_hi ? qend : q_hi;
  }
  const int ntile = qend > qbeg ? (qend - qbeg + 63) / 64 : 0;
  const int total = ntile * p.group;
  const float c = p.scale * LOG2E;

  // tile DMA: 16 one-KiB pieces per image over 8 waves = 2 per wave per image (piece = 2*wave8 + i: rows 8*wave8 + 4*i ..)
  uint32_t voff_q[2], voff_d[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row_ = 8 * wave8 + 4 * i + (lane >> 4);
    const int ch_ = (lane & 15) ^ (((row_ & 3) << 2) | ((row_ >> 2) & 3));
    voff_q[i] = (uint32_t)((row_ * p.q_st + ch_ * 8) * (int64_t)sizeof(e));
    voff_d[i] = (uint32_t)((row_ * p.o_st + ch_ * 8) * (int64_t)sizeof(e));
  }
#define KV2_DMA(HG, TI, B)                                                                                 \
  { const int hq_ = __builtin_amdgcn_readfirstlane(kvh * p.group + (HG));                                   \
    const int row0_ = qbeg + 64 * (TI) - p.q_offset;                                                       \
    char* base_ = smem + (B) * KV2_BUF;                                                                    \
    if (wave8 < 2) { int qr_ = row0_ + lane; qr_ = qr_ < p.Tq ? qr_ : p.Tq - 1;   /* wave 0: lse[64], wave 1: delta[64] */ \
      const float* src_ = (wave8 == 0 ? p.lse_r : p.delta) + (int64_t)hq_ * p.Tq;                          \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_ + qr_),        \
                                       (__attribute__((address_space(3))) void*)(base_ + 2 * TILE_BYTES + wave8 * 256), 4, 0, 0); } \
    const char* qb_ = reinterpret_cast<const char*>(p.q) + ((int64_t)hq_ * p.q_sh + (int64_t)row0_ * p.q_st) * (int64_t)sizeof(e);    \
    const char* db_ = reinterpret_cast<const char*>(p.dout) + ((int64_t)hq_ * p.o_sh + (int64_t)row0_ * p.o_st) * (int64_t)sizeof(e); \
    if (row0_ + 64 <= p.Tq) {                                                                              \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                   \
        uint32_t oq_ = voff_q[i_], od_ = voff_d[i_];                                                       \
        asm volatile("" : "+v"(oq_), "+v"(od_));                                                           \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qb_ + oq_),       \
                                         (__attribute__((address_space(3))) void*)(base_ + (wave8 * 2 + i_) * 1024), 16, 0, 0);              \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db_ + od_),       \
                                         (__attribute__((address_space(3))) void*)(base_ + TILE_BYTES + (wave8 * 2 + i_) * 1024), 16, 0, 0); } \
    } else {                                       /* ragged last tile of the tensor: clamp the row per lane */ \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                   \
        const int row_ = 8 * wave8 + 4 * i_ + (lane >> 4);                                                 \
        const int ch_ = (lane & 15) ^ (((row_ & 3) << 2) | ((row_ >> 2) & 3));                              \
        const int rr_ = row0_ + row_ < p.Tq ? row_ : p.Tq - 1 - row0_;                                     \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qb_ + ((int64_t)rr_ * p.q_st + ch_ * 8) * (int64_t)sizeof(e)), \
                                         (__attribute__((address_space(3))) void*)(base_ + (wave8 * 2 + i_) * 1024), 16, 0, 0);              \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db_ + ((int64_t)rr_ * p.o_st + ch_ * 8) * (int64_t)sizeof(e)), \
                                         (__attribute__((address_space(3))) void*)(base_ + TILE_BYTES + (wave8 * 2 + i_) * 1024), 16, 0, 0); } } }

  {
    int hg_c = 0, ti_c = 0;
    if (total > 0) KV2_DMA(hg_c, ti_c, 0)
    __syncthreads();
    int cur = 0;
    for (int idx = 0; idx < total; ++idx) {
      const int ti = ti_c;
      ti_c += 1;
      if (ti_c >= ntile) { ti_c = 0; ++hg_c; }
      if (idx + 1 < total) KV2_DMA(hg_c, ti_c, cur ^ 1)
      const float* lse_s = reinterpret_cast<const float*>(smem + cur * KV2_BUF + 2 * TILE_BYTES); const float* del_s = lse_s + 64;
      const int qi0 = qbeg + 64 * ti + 32 * grp;                       // packed index of this group's first row
      const bool full = (qbeg + 64 * ti >= k0 + KT - 1) && (qbeg + 64 * ti + 63 < se_min);   // workgroup-uniform: no mask needed
      const int sb = cur * KV2_BUF + grp * (32 * 256);
      int ar[8], at[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { ar[j] = offs.row[j] + sb; at[j] = offs.tr[j] + sb; }
      f32x16 S, DP;
#pragma unroll
      for (int g = 0; g < 16; ++g) { S[g] = 0.f; DP[g] = 0.f; }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const v8 aq = *reinterpret_cast<const v8*>(smem + ar[s]);
        const v8 ad = *reinterpret_cast<const v8*>(smem + ar[s] + TILE_BYTES);
        const v8 kfs = (DTA_ABL & 512) ? aq : *reinterpret_cast<const v8*>(kvs + s * 1024 + lane * 16);
        const v8 vfs = (DTA_ABL & 512) ? ad : *reinterpret_cast<const v8*>(kvs + 8192 + s * 1024 + lane * 16);
        S = T::mma(aq, kfs, S); DP = T::mma(ad, vfs, DP);
      }
      float nl[16], dl[16];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int ql = 32 * grp + 8 * gq + 4 * h;
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + ql);
        const float4 d4 = *reinterpret_cast<const float4*>(del_s + ql);
        nl[4 * gq] = l4.x; nl[4 * gq + 1] = l4.y; nl[4 * gq + 2] = l4.z; nl[4 * gq + 3] = l4.w;
        dl[4 * gq] = d4.x; dl[4 * gq + 1] = d4.y; dl[4 * gq + 2] = d4.z; dl[4 * gq + 3] = d4.w;
      }
      if (full) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const float pv = fast_exp2(__builtin_fmaf(S[g], c, -nl[g]));
          S[g] = pv;
          DP[g] = pv * (DP[g] - dl[g]);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int qi = qi0 + 8 * (g >> 2) + 4 * h + (g & 3);
          const bool ok = (kidx <= qi) && (qi < se_l);
          const float pv = ok ? fast_exp2(__builtin_fmaf(S[g], c, -nl[g])) : 0.f;
          S[g] = pv;
          DP[g] = pv * (DP[g] - dl[g]);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const v8 pb = pack_half<DT>(S, s2), sbf = pack_half<DT>(DP, s2);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const v8 adt = tr_pair<v8>(smem + at[db] + TILE_BYTES + 4096 * s2, smem + at[4 + db] + TILE_BYTES + 4096 * s2);
          const v8 aqt = tr_pair<v8>(smem + at[db] + 4096 * s2, smem + at[4 + db] + 4096 * s2);
          DV[db] = T::mma(adt, pb, DV[db]); DK[db] = T::mma(aqt, sbf, DK[db]);
        }
      }
      if (!(DTA_ABL & 256)) __syncthreads();
      cur ^= 1;
    }
  }
#undef KV2_DMA
  {
    // group 1 hands its partial sums to group 0 through LDS, 32 accumulators (one d-block of dK and dV) at a time
    float* red = reinterpret_cast<float*>(smem_all);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      if (grp == 1) {
#pragma unroll
        for (int g = 0; g < 16; ++g) { red[g * 256 + tid] = DK[db][g]; red[(16 + g) * 256 + tid] = DV[db][g]; }
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int g = 0; g < 16; ++g) { DK[db][g] += red[g * 256 + tid]; DV[db][g] += red[(16 + g) * 256 + tid]; }
      }
      __syncthreads();
    }
    if (grp == 1) return;
  }
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int g = 0; g < 16; ++g) DK[db][g] *= p.scale;
  const int kloc = wave * 32 + r;
  if (slab >= 0) {
    float* ws = p.dkv_ws + ((int64_t)slab * p.Hkv + kvh) * (2 * KT * 128) + (int64_t)kloc * 128;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d = 32 * db + 8 * gq + 4 * h;
        *reinterpret_cast<float4*>(ws + d) = make_float4(DK[db][4 * gq], DK[db][4 * gq + 1], DK[db][4 * gq + 2], DK[db][4 * gq + 3]);
        *reinterpret_cast<float4*>(ws + KT * 128 + d) = make_float4(DV[db][4 * gq], DV[db][4 * gq + 1], DV[db][4 * gq + 2], DV[db][4 * gq + 3]);
      }
  } else if (kidx < p.Tk) {
    e* dkp = reinterpret_cast<e*>(p.dk) + (int64_t)kidx * p.dkv_st + (int64_t)kvh * p.dkv_sh;
    e* dvp = reinterpret_cast<e*>(p.dv) + (int64_t)kidx * p.dkv_st + (int64_t)kvh * p.dkv_sh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d = 32 * db + 8 * gq + 4 * h;
        v4 wk, wv;
        if (p.accumulate) {
          const v4 ok_ = *reinterpret_cast<const v4*>(dkp + d); const v4 ov_ = *reinterpret_cast<const v4*>(dvp + d);
#pragma unroll
          for (int j = 0; j < 4; ++j) { wk[j] = (e)(DK[db][4 * gq + j] + (float)ok_[j]); wv[j] = (e)(DV[db][4 * gq + j] + (float)ov_[j]); }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) { wk[j] = (e)DK[db][4 * gq + j]; wv[j] = (e)DV[db][4 * gq + j]; }
        }
        *reinterpret_cast<v4*>(dkp + d) = wk;
        *reinterpret_cast<v4*>(dvp + d) = wv;
      }
  }
}

// Sums the fp32 slabs of every split key tile in a fixed order and writes dK/dV (bitwise reproducible).
// dkv_splits[s] = {key tile, first slab, number of slabs, 0}.
constexpr int FIN_SPLIT = 8;          // blockIdx.y: each (split key tile, kv head) is summed by 8 workgroups — the sums are load-latency bound
template <int DT>
__global__ __launch_bounds__(256) void tree_attn_bwd_dkv_finalize_kernel(AttnParams p) {
  using e = typename Ty<DT>::e; using v4 = typename Ty<DT>::v4;
  const int KT = p.ktile;
  const int kvh = blockIdx.x % p.Hkv, sp = blockIdx.x / p.Hkv;
  const int kt = p.dkv_splits[4 * sp], first = p.dkv_splits[4 * sp + 1], n = p.dkv_splits[4 * sp + 2];
  const int per = 2 * KT * 32 / FIN_SPLIT;                                // float4 indices per workgroup
  const float* ws0 = p.dkv_ws + ((int64_t)first * p.Hkv + kvh) * (2 * KT * 128);
  const int64_t slab_st = (int64_t)p.Hkv * (2 * KT * 128);
  for (int i = blockIdx.y * per + threadIdx.x; i < (blockIdx.y + 1) * per; i += 256) {   // float4 index inside a slab
    const int which = i / (KT * 32), rem = i - which * KT * 32;
    const int key = rem >> 5, d = (rem & 31) << 2;
    const int kidx = kt * KT + key;
    if (kidx >= p.Tk) continue;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int j = 0;
    for (; j + 4 <= n; j += 4) {                                          // four loads in flight, summed in slab order
      const float4 a0 = *reinterpret_cast<const float4*>(ws0 + (j + 0) * slab_st + (int64_t)i * 4);
      const float4 a1 = *reinterpret_cast<const float4*>(ws0 + (j + 1) * slab_st + (int64_t)i * 4);
      const float4 a2 = *reinterpret_cast<const float4*>(ws0 + (j + 2) * slab_st + (int64_t)i * 4);
      const float4 a3 = *reinterpret_cast<const float4*>(ws0 + (j + 3) * slab_st + (int64_t)i * 4);
      acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w;
      acc.x += a1.x; acc.y += a1.y; acc.z += a1.z; acc.w += a1.w;
      acc.x += a2.x; acc.y += a2.y; acc.z += a2.z; acc.w += a2.w;
      acc.x += a3.x; acc.y += a3.y; acc.z += a3.z; acc.w += a3.w;
    }
    for (; j < n; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(ws0 + j * slab_st + (int64_t)i * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    e* out = reinterpret_cast<e*>(which ? p.dv : p.dk) + (int64_t)kidx * p.dkv_st + (int64_t)kvh * p.dkv_sh + d;
    if (p.accumulate) { const v4 o = *reinterpret_cast<const v4*>(out); acc.x += (float)o[0]; acc.y += (float)o[1]; acc.z += (float)o[2]; acc.w += (float)o[3]; }
    v4 w; w[0] = (e)acc.x; w[1] = (e)acc.y; w[2] = (e)acc.z; w[3] = (e)acc.w;
    *reinterpret_cast<v4*>(out) = w;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int dta_tree_attn_fwd_ex(const void* q, const void* k, const void* v, void* out, float* lse,
                                    const int32_t* subtree_end, const int32_t* run_ptr, const int32_t* runs,
                                    int32_t Tq, int32_t Tk, int32_t q_offset, int32_t Hq, int32_t Hkv, int32_t head_dim,
                                    int64_t q_st, int64_t q_sh, int64_t kv_st, int64_t kv_sh, int64_t v_st, int64_t v_sh, int64_t o_st, int64_t o_sh,
                                    float scale, int32_t dtype, void* stream) {
  if (!q || !k || !v || !out || !lse || Tq <= 0 || Tk <= 0 || Hq <= 0 || Hkv <= 0 || q_offset < 0) return DTA_EINVAL;
  if ((runs == nullptr) != (run_ptr == nullptr)) return DTA_EINVAL;
  if (head_dim != 128 || Hq % Hkv != 0 || (dtype != DTA_BF16 && dtype != DTA_F16)) return DTA_EUNSUPPORTED;
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out) || (q_st | q_sh | kv_st | kv_sh | v_st | v_sh | o_st | o_sh) % 8 != 0) return DTA_EALIGN;
  // the tile DMA addresses a 64-row tile as scalar base + 32-bit lane offset: token strides must keep 64 rows inside 4 GiB
  if (kv_st < 0 || v_st < 0 || kv_st > (1 << 24) || v_st > (1 << 24)) return DTA_EUNSUPPORTED;
  AttnParams p{};
  p.q = q; p.k = k; p.v = v; p.out = out; p.lse_w = lse; p.subtree_end = subtree_end; p.run_ptr = run_ptr; p.runs = runs;
  p.Tq = Tq; p.Tk = Tk; p.q_offset = q_offset; p.Hq = Hq; p.Hkv = Hkv; p.group = Hq / Hkv;
  p.q_st = q_st; p.q_sh = q_sh; p.kv_st = kv_st; p.kv_sh = kv_sh; p.v_st = v_st; p.v_sh = v_sh; p.o_st = o_st; p.o_sh = o_sh; p.scale = scale;
  const int nqt = (Tq + DTA_QTILE - 1) / DTA_QTILE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();   // drop a stale error of an earlier, unrelated runtime call
  if (p.group % 2 == 0 && !(DTA_ABL & 16)) {      // two query heads of a kv group share the staged K/V tiles
    dim3 grid(nqt * Hq / 2), block(512);
    if (dtype == DTA_BF16) hipLaunchKernelGGL((tree_attn_fwd_kernel<DTA_BF16, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((tree_attn_fwd_kernel<DTA_F16, 2>), grid, block, 0, st, p);
  } else {
    dim3 grid(nqt * Hq), block(256);
    if (dtype == DTA_BF16) hipLaunchKernelGGL((tree_attn_fwd_kernel<DTA_BF16, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((tree_attn_fwd_kernel<DTA_F16, 1>), grid, block, 0, st, p);
  }
  return hipGetLastError() == hipSuccess ? DTA_OK : DTA_ELAUNCH;
}

extern "C" int dta_tree_attn_bwd_ex(const void* q, const void* k, const void* v, const void* out, const void* dout,
                                    const float* lse, float* delta, void* dq, void* dk, void* dv,
                                    const int32_t* subtree_end, const int32_t* run_ptr, const int32_t* runs,
                                    const int32_t* ktile_qend,
                                    int32_t Tq, int32_t Tk, int32_t q_offset, int32_t Hq, int32_t Hkv, int32_t head_dim,
                                    int64_t q_st, int64_t q_sh, int64_t kv_st, int64_t kv_sh, int64_t v_st, int64_t v_sh, int64_t o_st, int64_t o_sh,
                                    int64_t dq_st, int64_t dq_sh, int64_t dkv_st, int64_t dkv_sh,
                                    float scale, int32_t dtype, int32_t accumulate, int32_t which,
                                    const int32_t* dkv_units, int32_t n_units, const int32_t* dkv_splits, int32_t n_splits, float* dkv_ws,
                                    void* stream) {
  if (!q || !k || !v || !out || !dout || !lse || !delta || !dq || !dk || !dv || Tq <= 0 || Tk <= 0 || Hq <= 0 || Hkv <= 0 || q_offset < 0) return DTA_EINVAL;
  if ((runs == nullptr) != (run_ptr == nullptr)) return DTA_EINVAL;
  if (dkv_units && (n_units <= 0 || n_splits < 0 || (n_splits > 0 && (!dkv_splits || !dkv_ws)))) return DTA_EINVAL;
  if (head_dim != 128 || Hq % Hkv != 0 || (dtype != DTA_BF16 && dtype != DTA_F16)) return DTA_EUNSUPPORTED;
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out) || !aligned16(dout) || !aligned16(dq) || !aligned16(dk) || !aligned16(dv) ||
      (q_st | q_sh | kv_st | kv_sh | v_st | v_sh | o_st | o_sh | dq_st | dq_sh | dkv_st | dkv_sh) % 8 != 0) return DTA_EALIGN;
  if (q_st < 0 || o_st < 0 || q_st > (1 << 24) || o_st > (1 << 24)) return DTA_EUNSUPPORTED;   // 64-row tile = scalar base + 32-bit lane offset
  AttnParams p{};
  p.q = q; p.k = k; p.v = v; p.o = out; p.dout = dout; p.lse_r = lse; p.delta = delta; p.dq = dq; p.dk = dk; p.dv = dv;
  p.subtree_end = subtree_end; p.run_ptr = run_ptr; p.runs = runs; p.ktile_qend = ktile_qend;
  p.dkv_units = dkv_units; p.dkv_splits = dkv_splits; p.dkv_ws = dkv_ws;
  p.Tq = Tq; p.Tk = Tk; p.q_offset = q_offset; p.Hq = Hq; p.Hkv = Hkv; p.group = Hq / Hkv;
  p.q_st = q_st; p.q_sh = q_sh; p.kv_st = kv_st; p.kv_sh = kv_sh; p.v_st = v_st; p.v_sh = v_sh; p.o_st = o_st; p.o_sh = o_sh;
  p.dq_st = dq_st; p.dq_sh = dq_sh; p.dkv_st = dkv_st; p.dkv_sh = dkv_sh; p.scale = scale; p.accumulate = accumulate;
  const int nqt = (Tq + DTA_QTILE - 1) / DTA_QTILE;
  // KB = 2 (64 keys per wave, every LDS fragment feeding two MFMAs) is written but needs > 512 registers
  // with K and V fragments resident; it stays out of the build until their staging moves to LDS/DMA.
  p.ktile = DTA_KTILE;
  const int nkt = (Tk + p.ktile - 1) / p.ktile;
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();   // drop a stale error of an earlier, unrelated runtime call
  if ((which & 7) == 0) return DTA_EINVAL;
  const bool fin = ((which & 2) && !(which & 8)) || (which & 4);     // slab finalize: with the dK/dV launch unless bit3, or alone (bit2)
  // NG = 2 (two wave groups, two waves per SIMD) was measured at 0.52x the speed of NG = 1 on the tau2 trie: 128
  // accumulators + 64 K/V fragment registers do not fit 256 registers per wave (69 spills).  Not instantiated.
  const int ndkv = dkv_units ? n_units : nkt;
  const bool pair = p.group % 2 == 0;
  const dim3 gq(pair ? nqt * Hq / 2 : nqt * Hq), bq(pair ? 512 : 256);
  if (dtype == DTA_BF16) {
    if (which & 1) { if (pair) hipLaunchKernelGGL((tree_attn_bwd_dq_kernel<DTA_BF16, 2>), gq, bq, 0, st, p); else hipLaunchKernelGGL((tree_attn_bwd_dq_kernel<DTA_BF16, 1>), gq, bq, 0, st, p); }
    if (which & 2) { if (!(DTA_ABL & 128)) hipLaunchKernelGGL((tree_attn_bwd_dkv2_kernel<DTA_BF16>), dim3(ndkv * Hkv), dim3(512), 0, st, p);
                     else hipLaunchKernelGGL((tree_attn_bwd_dkv_kernel<DTA_BF16, DTA_KTILE / 128, 1>), dim3(ndkv * Hkv), dim3(256), 0, st, p); }
    if (fin && dkv_units && n_splits > 0) hipLaunchKernelGGL(tree_attn_bwd_dkv_finalize_kernel<DTA_BF16>, dim3(n_splits * Hkv, FIN_SPLIT), dim3(256), 0, st, p);
  } else {
    if (which & 1) { if (pair) hipLaunchKernelGGL((tree_attn_bwd_dq_kernel<DTA_F16, 2>), gq, bq, 0, st, p); else hipLaunchKernelGGL((tree_attn_bwd_dq_kernel<DTA_F16, 1>), gq, bq, 0, st, p); }
    if (which & 2) { if (!(DTA_ABL & 128)) hipLaunchKernelGGL((tree_attn_bwd_dkv2_kernel<DTA_F16>), dim3(ndkv * Hkv), dim3(512), 0, st, p);
                     else hipLaunchKernelGGL((tree_attn_bwd_dkv_kernel<DTA_F16, DTA_KTILE / 128, 1>), dim3(ndkv * Hkv), dim3(256), 0, st, p); }
    if (fin && dkv_units && n_splits > 0) hipLaunchKernelGGL(tree_attn_bwd_dkv_finalize_kernel<DTA_F16>, dim3(n_splits * Hkv, FIN_SPLIT), dim3(256), 0, st, p);
  }
  return hipGetLastError() == hipSuccess ? DTA_OK : DTA_ELAUNCH;
}

// Token-major convenience forms declared in dta.h: head stride = 128 elements.
extern "C" int dta_tree_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
                                 const int32_t* subtree_end, const int32_t* run_ptr, const int32_t* runs,
                                 int32_t Tq, int32_t Tk, int32_t q_offset, int32_t Hq, int32_t Hkv, int32_t head_dim,
                                 int64_t q_stride_t, int64_t kv_stride_t, int64_t o_stride_t,
                                 float scale, int32_t dtype, void* stream) {
  return dta_tree_attn_fwd_ex(q, k, v, out, lse, subtree_end, run_ptr, runs, Tq, Tk, q_offset, Hq, Hkv, head_dim,
                              q_stride_t, 128, kv_stride_t, 128, kv_stride_t, 128, o_stride_t, 128, scale, dtype, stream);
}

extern "C" int dta_tree_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                                 const float* lse, float* delta, void* dq, void* dk, void* dv,
                                 const int32_t* subtree_end, const int32_t* run_ptr, const int32_t* runs,
                                 const int32_t* ktile_qend,
                                 int32_t Tq, int32_t Tk, int32_t q_offset, int32_t Hq, int32_t Hkv, int32_t head_dim,
                                 int64_t q_stride_t, int64_t kv_stride_t, int64_t o_stride_t,
                                 int64_t dq_stride_t, int64_t dkv_stride_t,
                                 float scale, int32_t dtype, int32_t accumulate, void* stream) {
  return dta_tree_attn_bwd_ex(q, k, v, out, dout, lse, delta, dq, dk, dv, subtree_end, run_ptr, runs, ktile_qend,
                              Tq, Tk, q_offset, Hq, Hkv, head_dim, q_stride_t, 128, kv_stride_t, 128, kv_stride_t, 128, o_stride_t, 128,
                              dq_stride_t, 128, dkv_stride_t, 128, scale, dtype, accumulate, 3, nullptr, 0, nullptr, 0, nullptr, stream);
}
