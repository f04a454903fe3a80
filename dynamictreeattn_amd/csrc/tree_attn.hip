// Tree attention forward / backward for gfx950 (MI355X, CDNA4).  head_dim = 128, bf16 or f16.
//
// One kernel family serves both forms of the reference's "node attends to its ancestor path":
//   * packed trie (DFS pre-order): key s visible to query t  <=>  s <= t < subtree_end[s]
//   * stack form (tree_training_engine.py:171-186): subtree_end == NULL, q_offset = start
//
// Tiling (wave64, v_mfma_f32_32x32x16, two waves per SIMD everywhere):
//   fwd / dQ : workgroup = 8 waves = 128 query rows x the TWO query heads of one kv group (4 waves = 128 rows of one
//              head when the group is odd); each wave owns 32 rows with the QUERY ON THE MFMA LANE (S^T = K.Q^T), so
//              the softmax row statistics are lane-local and the S^T accumulator is directly the B operand of
//              O^T += V^T.P^T / dQ^T += K^T.dS^T.  Both heads share the staged 64-key K/V tiles.
//   dK/dV    : workgroup = 8 waves = 128 keys of one kv head with the KEY ON THE LANE (S = Q.K^T): the two 4-wave groups
//              own the same keys and split every 64-row query tile; dK^T/dV^T live in 128 accumulator registers per wave
//              across the whole query sweep (all query heads of the GQA group); the K/V fragments sit in LDS in fragment
//              order.  Heavy key tiles are cut into split-Q work units whose fp32 slabs a finalize launch sums in order:
//              no atomics, bitwise reproducible.
//   Tiles (K/V for fwd, Q/dO for dK/dV) go global -> LDS by LDS-DMA issued from INLINE ASM (dma_* below) into one
//   XOR-swizzled 256-B-row image that serves BOTH row reads (ds_read_b128) and transposed reads (ds_read_b64_tr_b16);
//   the dQ kernel stages through registers (issue early, write late).
//
// Why inline asm for the DMA: with the builtin, hipcc (ROCm 7.2) waits `vmcnt(0)` for the in-flight prefetch of the
// NEXT tile in front of LDS reads of the CURRENT one (before the first ds_read when the kernel has a second __shared__
// object, before the first transposed / float4 read otherwise) - the prefetch was exposed on every tile.  An asm DMA is
// outside hipcc's bookkeeping (cdna guide 5.7): the only wait is our own `s_waitcnt vmcnt(0)` in front of the
// tile-end barrier, so a tile's DMA has the whole compute phase of the previous tile to land.
//
// Lane maps used here were verified on hardware by tests/micro/mfma_layout_probe.hip.
// The round-1 ablation switches (-DDTA_ABL) and the retired 4-wave dK/dV kernel live in scripts/diag/ (not built).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "dta_common.h"

// Diagnostic build switch (-DDTA_PRIO_HALF=1): static priority for the second-dispatched half of an 8-wave workgroup (waves 4-7 lose every
// issue arbitration by age; MI355X_MICROARCH.md "Two waves per SIMD", item 4).  Measured in round 3: see DESIGN.md §9c.
#if defined(DTA_PRIO_HALF) && DTA_PRIO_HALF
#define DTA_PRIO_YOUNGER_HALF(W) if ((W) >= 4) __builtin_amdgcn_s_setprio(1);
#else
#define DTA_PRIO_YOUNGER_HALF(W)
#endif

// Diagnostic build switch (-DDTA_STAMP=1): in-kernel s_memtime stamps at the segment boundaries of the forward's tile loop, summed per
// wave in scalar registers and written to a debug buffer of their own (cdna_hip_programming.md §7 "In-kernel stamps"); scripts/fwd_stamps.py
// reads the SHARES.  No stamp executes in the product build.
#if defined(DTA_STAMP) && DTA_STAMP
__device__ unsigned long long* dta_stamp_buf = nullptr;
extern "C" int dta_debug_set_stamp_buffer(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(dta_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -4; }
#define DTA_STAMP_DECL unsigned long long st_prev_, st_sum_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long st_tiles_ = 0; \
  __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev_) :: "memory"); __builtin_amdgcn_sched_barrier(0);
#define DTA_STAMP_AT(K) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  __builtin_amdgcn_sched_barrier(0); st_sum_[K] += t_ - st_prev_; st_prev_ = t_; }
#define DTA_STAMP_TILE ++st_tiles_;
#define DTA_STAMP_STORE if (dta_stamp_buf && lane == 0) { unsigned long long* o_ = dta_stamp_buf + ((size_t)blockIdx.x * 8 + wave) * 8; \
  for (int k_ = 0; k_ < 6; ++k_) o_[k_] = st_sum_[k_]; o_[6] = st_tiles_; o_[7] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) /* HW_ID: wave / simd / cu ids */; }
#else
#define DTA_STAMP_DECL
#define DTA_STAMP_AT(K)
#define DTA_STAMP_TILE
#define DTA_STAMP_STORE
#endif

// Waves of a forward / dQ workgroup that ISSUE the K/V tile DMA.  In an 8-wave workgroup the second-dispatched half (waves 4-7) loses every
// issue arbitration on its SIMD and is the tile's critical path, while the first half waits a quarter of the tile at the barrier (stamps:
// DESIGN.md §9c): -DDTA_DMA_OLDER_HALF=1 (diagnostic) lets waves 0-3 issue all 32 pieces (8 each) and the critical half none.  Measured:
// the younger half's time moves from its DMA segment into its softmax / PV segments, the tile takes as long as before (zero-sum).
#if defined(DTA_DMA_OLDER_HALF) && DTA_DMA_OLDER_HALF
#define DTA_DMA_WAVES(HPB) 4
#define DTA_DMA_GUARD(NW) if (wave < (NW))
#else
#define DTA_DMA_WAVES(HPB) (4 * (HPB))
#define DTA_DMA_GUARD(NW)
#endif

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
  int32_t hgroups, head0;          // forward / dQ launch: workgroups per (query tile, kv head) and the first query head (inside a kv group) they cover
  int64_t q_st, q_sh, kv_st, kv_sh, v_st, v_sh, o_st, o_sh, dq_st, dq_sh, dkv_st, dkv_sh;
  float scale; int32_t accumulate; int32_t ktile;
};

#ifndef DTA_FWD_FORM_DEFAULT
#define DTA_FWD_FORM_DEFAULT 1
#endif
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

// Deferred reference maximum of the forward (log2 domain): the running reference follows a tile's row maximum only when that exceeds it by
// more than this, so P <= 2^THR (fp32 sums; bf16 P keeps its relative precision) and the rescale of O - taken on nine tiles in ten with
// THR = 0, because ANY of a wave's rows triggers it - becomes rare.  -DDTA_FWD_THR=0 restores the exact-maximum form.
#ifndef DTA_FWD_THR
#define DTA_FWD_THR 4.0f
#endif
constexpr float FWD_THR = DTA_FWD_THR;
constexpr int SE_BYTES = 256;                                       // 64 x int32 subtree_end of the staged keys
constexpr int QK_LDS = 2 * (2 * TILE_BYTES + SE_BYTES);            // double-buffered {K image, V image, se}

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float half_max(float x) {      // max(x, value of the lane 32 away) by v_permlane32_swap (no LDS round trip)
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

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

// ---- LDS-DMA from inline asm (see the file header for why) ---------------------------------------------------------
// A wave instruction lands 64 x 16 B = 1 KiB = 4 image rows lane-linearly at M0, so the image's XOR swizzle goes on the
// per-lane SOURCE chunk.  The global address is <scalar base> + <32-bit per-lane byte offset>: per tile only the base
// moves.  `s_nop 4` covers a base that was just produced by v_readfirstlane (VALU-written SGPR -> VMEM, 5 wait states),
// `s_nop 0` the M0 write -> LDS-DMA hazard.  hipcc does not count these loads: DMA_WAIT() before the barrier that
// publishes the tile is the only thing that orders them.
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(p);
}
// pieces {0, 1} of image A (at lds, lds + 1 KiB) and of image B (at lds + TILE_BYTES, + 1 KiB)
__device__ __forceinline__ void dma_pair2(uint32_t oa0, uint32_t oa1, const void* ba, uint32_t ob0, uint32_t ob1, const void* bb, uint32_t lds) {
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 m0, %6\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %4\n\t"
      "s_add_u32 m0, %6, 0x4000\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %5\n\t"
      "s_add_u32 m0, %6, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %4\n\t"
      "s_add_u32 m0, %6, 0x4400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, %5"
      :: "v"(oa0), "v"(oa1), "v"(ob0), "v"(ob1), "s"(ba), "s"(bb), "s"(lds) : "memory", "scc");
}
// 64 dwords (row constants of a tile: subtree_end, lse, delta)
__device__ __forceinline__ void dma_dword(uint32_t off, const void* base, uint32_t lds) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(off), "s"(base), "s"(lds) : "memory");
}
#define DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// Byte offset of this lane's 16-B chunk of image row `row` (rows of `stride` elements of `esz` bytes), swizzled
__device__ __forceinline__ uint32_t dma_src_off(int row, int img_row, int lane, int64_t stride, int esz) {
  const int ch = (lane & 15) ^ (((img_row & 3) << 2) | ((img_row >> 2) & 3));
  return (uint32_t)((row * stride + ch * 8) * (int64_t)esz);
}

// K/V tile pair of the forward: NW waves move the 16 + 16 pieces; wave w owns pieces (16/NW)*w .. of both images.
// subtree_end of the 64 keys goes by 4-byte DMA from wave 0; keys at or beyond the run end are excluded by the
// caller's `k <= min(q, kend-1)` test, not by a sentinel.
#define DTA_KV_OFFSETS(NW)                                                                                 \
  uint32_t voff_k[16 / (NW)], voff_v[16 / (NW)];                                                           \
  _Pragma("unroll") for (int i_ = 0; i_ < 16 / (NW); ++i_) {                                               \
    const int row_ = 4 * (wave * (16 / (NW)) + i_) + (lane >> 4);                                          \
    voff_k[i_] = dma_src_off(row_, row_, lane, p.kv_st, sizeof(e));                                        \
    voff_v[i_] = dma_src_off(row_, row_, lane, p.v_st, sizeof(e)); }
#define DTA_KV_DMA(BASE, K0, NW)                                                                           \
  DTA_DMA_GUARD(NW) { char* base_ = (BASE); const int k0_ = (K0);                                          \
    if (wave == 0) {                                                                                       \
      if (p.subtree_end) { int ki_ = k0_ + lane; ki_ = ki_ < p.Tk ? ki_ : p.Tk - 1;                        \
        dma_dword((uint32_t)ki_ * 4u, p.subtree_end, lds_addr(base_ + 2 * TILE_BYTES)); }                  \
      else reinterpret_cast<int*>(base_ + 2 * TILE_BYTES)[lane] = 0x7fffffff; }                            \
    const char* kb_ = reinterpret_cast<const char*>(kbase) + (int64_t)k0_ * p.kv_st * (int64_t)sizeof(e);  \
    const char* vb_ = reinterpret_cast<const char*>(vbase) + (int64_t)k0_ * p.v_st * (int64_t)sizeof(e);   \
    uint32_t ok_[16 / (NW)], ov_[16 / (NW)];                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 16 / (NW); ++i_) { ok_[i_] = voff_k[i_]; ov_[i_] = voff_v[i_]; } \
    if (k0_ + 64 > p.Tk) {                         /* ragged last tile of the tensor: clamp the row per lane */ \
      _Pragma("unroll") for (int i_ = 0; i_ < 16 / (NW); ++i_) {                                           \
        const int row_ = 4 * (wave * (16 / (NW)) + i_) + (lane >> 4);                                      \
        const int rr_ = k0_ + row_ < p.Tk ? row_ : p.Tk - 1 - k0_;                                         \
        ok_[i_] = dma_src_off(rr_, row_, lane, p.kv_st, sizeof(e)); ov_[i_] = dma_src_off(rr_, row_, lane, p.v_st, sizeof(e)); } } \
    _Pragma("unroll") for (int i_ = 0; i_ < 16 / (NW); i_ += 2)                                            \
      dma_pair2(ok_[i_], ok_[i_ + 1], kb_, ov_[i_], ov_[i_ + 1], vb_, lds_addr(base_ + (wave * (16 / (NW)) + i_) * 1024)); }

// The same tile DMA in two steps, so that a one-wave-per-SIMD kernel can issue the pieces BETWEEN its MFMAs instead of in one exposed burst
// (8 pieces cost a wave ~900 cycles of issue): DTA_KV_DMA_PREP declares the bases / offsets (and sends the subtree_end row), DTA_KV_DMA_PAIR(i)
// issues piece pair i (i = 0 .. 16/NW/2 - 1) of both images.
#define DTA_KV_DMA_PREP(BASE, K0, NW, COND)                                                                    \
  char* dbase_ = (BASE); const int dk0_ = (K0);                                                            \
  if (wave == 0 && (COND)) {                                                                                      \
    if (p.subtree_end) { int ki_ = dk0_ + lane; ki_ = ki_ < p.Tk ? ki_ : p.Tk - 1;                         \
      dma_dword((uint32_t)ki_ * 4u, p.subtree_end, lds_addr(dbase_ + 2 * TILE_BYTES)); }                   \
    else reinterpret_cast<int*>(dbase_ + 2 * TILE_BYTES)[lane] = 0x7fffffff; }                             \
  const char* dkb_ = reinterpret_cast<const char*>(kbase) + (int64_t)dk0_ * p.kv_st * (int64_t)sizeof(e);  \
  const char* dvb_ = reinterpret_cast<const char*>(vbase) + (int64_t)dk0_ * p.v_st * (int64_t)sizeof(e);   \
  uint32_t dok_[16 / (NW)], dov_[16 / (NW)];                                                               \
  _Pragma("unroll") for (int i_ = 0; i_ < 16 / (NW); ++i_) { dok_[i_] = voff_k[i_]; dov_[i_] = voff_v[i_]; } \
  if (dk0_ + 64 > p.Tk) {                                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < 16 / (NW); ++i_) {                                             \
      const int row_ = 4 * (wave * (16 / (NW)) + i_) + (lane >> 4);                                        \
      const int rr_ = dk0_ + row_ < p.Tk ? row_ : p.Tk - 1 - dk0_;                                         \
      dok_[i_] = dma_src_off(rr_, row_, lane, p.kv_st, sizeof(e)); dov_[i_] = dma_src_off(rr_, row_, lane, p.v_st, sizeof(e)); } }
#define DTA_KV_DMA_PAIR(I, NW)                                                                             \
  dma_pair2(dok_[2 * (I)], dok_[2 * (I) + 1], dkb_, dov_[2 * (I)], dov_[2 * (I) + 1], dvb_, lds_addr(dbase_ + (wave * (16 / (NW)) + 2 * (I)) * 1024));

// The first DTA_V_PRELOAD_N V fragments (k-step by k-step) requested BEFORE the row maximum (they do not depend on the softmax; hipcc otherwise issues them right in
// front of the first PV MFMA, which then waits an LDS latency).  -DDTA_V_PRELOAD_N=0 disables.
#ifndef DTA_V_PRELOAD_N
#define DTA_V_PRELOAD_N 4
#endif
#if DTA_V_PRELOAD_N
#define DTA_V_PRELOAD v8 vpre_[DTA_V_PRELOAD_N]; _Pragma("unroll") for (int i_ = 0; i_ < DTA_V_PRELOAD_N; ++i_) { vpre_[i_] = tr_frag_o<v8>(Vs + 4096 * (i_ >> 2), offs, i_ & 3); asm volatile("" : "+v"(vpre_[i_])); }
#define DTA_V_FRAG(S4, DB) ((4 * (S4) + (DB) < DTA_V_PRELOAD_N) ? vpre_[4 * (S4) + (DB) < DTA_V_PRELOAD_N ? 4 * (S4) + (DB) : 0] : tr_frag_o<v8>(Vs + 4096 * (S4), offs, (DB)))
#else
#define DTA_V_PRELOAD
#define DTA_V_FRAG(S4, DB) tr_frag_o<v8>(Vs + 4096 * (S4), offs, (DB))
#endif

// The 16 score MFMAs of a tile.  Default: the two key blocks' chains INTERLEAVED with the fragment reads one k-step ahead (every MFMA's
// operand was requested two MFMAs earlier and consecutive MFMAs do not depend on each other); -DDTA_SCORE_ORDER=0: block after block
// (hipcc then waits for each of the first eight fragment reads right after issuing it).
#ifndef DTA_SCORE_ORDER
#define DTA_SCORE_ORDER 1
#endif
#if DTA_SCORE_ORDER
#define DTA_SCORE_MFMAS                                                                                    \
    { v8 ka_ = *reinterpret_cast<const v8*>(Ks + offs.row[0]), kb2_ = *reinterpret_cast<const v8*>(Ks + 8192 + offs.row[0]); \
      _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                                      \
        v8 na_ = ka_, nb_ = kb2_;                                                                          \
        if (s < 7) { na_ = *reinterpret_cast<const v8*>(Ks + offs.row[s + 1]); nb_ = *reinterpret_cast<const v8*>(Ks + 8192 + offs.row[s + 1]); } \
        X[0] = T::mma(ka_, qf[s], X[0]); X[1] = T::mma(kb2_, qf[s], X[1]);                                 \
        ka_ = na_; kb2_ = nb_;                                                                             \
      } }
#else
#define DTA_SCORE_MFMAS                                                                                    \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                       \
      _Pragma("unroll") for (int s = 0; s < 8; ++s)                                                        \
        X[kb] = T::mma(*reinterpret_cast<const v8*>(Ks + 8192 * kb + offs.row[s]), qf[s], X[kb]);
#endif

// =================================================================================================
// forward.  HPB = query heads of one kv group handled by a workgroup (waves 4*hb .. 4*hb+3 own head hb);
// they share the staged K/V tiles.  One barrier per 64-key tile, LDS double buffered, tile loop unrolled
// over the two buffers so that every LDS address is lane-offset + immediate.
// =================================================================================================
template <int DT, int HPB>
__global__ __launch_bounds__(256 * HPB, 2) void tree_attn_fwd_kernel(AttnParams p) {
  using T = Ty<DT>; using e = typename T::e; using v8 = typename T::v8; using v4 = typename T::v4;
  constexpr int NW = DTA_DMA_WAVES(HPB), BUF = 2 * TILE_BYTES + SE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[QK_LDS];

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform values live in SGPRs
  DTA_PRIO_YOUNGER_HALF(wave)
  const int hb = wave >> 2, rw = wave & 3;
  const int bid = blockIdx.x;
  const int hgroups = p.hgroups;
  const int kvh = bid % p.Hkv; const int rest = bid / p.Hkv; const int hgb = rest % hgroups;
  const int nqt = (p.Tq + DTA_QTILE - 1) / DTA_QTILE;
  const int qt = nqt - 1 - rest / hgroups;                          // deepest (heaviest) query tiles first
  const int hq = kvh * p.group + p.head0 + hgb * HPB + hb;
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
  // The Q fragments must be COMPLETE in hipcc's own book-keeping before the loop: it cannot see the asm DMA, but it does count the plain
  // global loads of Q, and with them still "pending" at the loop header it put `s_waitcnt vmcnt(7) .. vmcnt(0)` in front of the first eight
  // score MFMAs of the loop body - where vmcnt(0) also waits for the NEXT tile's DMA issued a few hundred cycles earlier.
#pragma unroll
  for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(qf[s]));
  DMA_WAIT(); __syncthreads();

  // one tile out of buffer BUFI (compile-time): prefetch the next tile into the other buffer, S^T, softmax, PV
#define FWD_TILE(BUFI)                                                                                     \
  {                                                                                                        \
    int nk0_ = 0, nkend_ = 0; bool nmask_ = false;                                                         \
    DTA_STAMP_AT(5) DTA_STAMP_TILE                                                                         \
    if (has_next) { nk0_ = it.k0; nkend_ = it.kend; nmask_ = it.masked(); DTA_KV_DMA(smem + (1 - (BUFI)) * BUF, it.k0, NW) } \
    DTA_STAMP_AT(0)                                                                                        \
    const char* Ks = smem + (BUFI) * BUF; const char* Vs = Ks + TILE_BYTES;                                \
    const int* se_s = reinterpret_cast<const int*>(Ks + 2 * TILE_BYTES);                                   \
    f32x16 X[2];                                                                                           \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                       \
      _Pragma("unroll") for (int g = 0; g < 16; ++g) X[kb][g] = 0.f;                                       \
    DTA_SCORE_MFMAS                                                                                        \
    DTA_STAMP_AT(1)                                                                                        \
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
    DTA_V_PRELOAD                                                                                          \
    float mx = max3(X[0][0], X[0][1], X[0][2]), mx1 = max3(X[1][0], X[1][1], X[1][2]);   /* two independent chains */ \
    _Pragma("unroll") for (int g = 3; g < 15; g += 2) { mx = max3(mx, X[0][g], X[0][g + 1]); mx1 = max3(mx1, X[1][g], X[1][g + 1]); } \
    mx = max3(mx, X[0][15], mx1);                                                                          \
    mx = half_max(fmaxf(mx, X[1][15]));        /* v_permlane32_swap: no LDS round trip (ds_bpermute + 6 address instructions before) */ \
    const float mc = mx * c;                                                                               \
    if (__any(mc > m + FWD_THR)) {             /* O is rescaled only when some row's maximum grew by more than the deferral threshold */ \
      const float mnew = fmaxf(m, mc);                                                                     \
      const float alpha = fast_exp2(m - mnew);                                                             \
      m = mnew; lsum *= alpha;                                                                             \
      _Pragma("unroll") for (int db = 0; db < 4; ++db)                                                     \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) O[db][g] *= alpha;                                  \
    }                                                                                                      \
    DTA_STAMP_AT(2)                                                                                        \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                       \
      _Pragma("unroll") for (int g = 0; g < 16; ++g) { const float pv = fast_exp2(__builtin_fmaf(X[kb][g], c, -m)); lsum += pv; X[kb][g] = pv; } \
    _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                                     \
      const v8 pb = pack_half<DT>(X[s4 >> 1], s4 & 1);                                                     \
      _Pragma("unroll") for (int db = 0; db < 4; ++db) O[db] = T::mma(DTA_V_FRAG(s4, db), pb, O[db]);      \
    }                                                                                                      \
    DTA_STAMP_AT(3)                                                                                        \
    DMA_WAIT(); __syncthreads();               /* the next tile has landed in every wave's view */          \
    DTA_STAMP_AT(4)                                                                                        \
    if (!has_next) break;                                                                                  \
    ck0 = nk0_; ckend = nkend_; cmask = nmask_;                                                            \
    has_next = it.advance();                                                                               \
  }
  DTA_STAMP_DECL
  while (true) {
    FWD_TILE(0)
    FWD_TILE(1)
  }
#undef FWD_TILE
  DTA_STAMP_STORE

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
// forward, 8 waves with the two head groups HALF A TILE APART (A/B form 4).  Same work split as tree_attn_fwd_kernel<DT, 2> (waves 0-3 =
// head 0, waves 4-7 = head 1 of the kv group, sharing the staged tiles, one barrier per tile), but the second group runs its tile body
// ROTATED: in the interval of tile j it first multiplies P(j-1) - kept packed in 16 registers across the barrier - into V(j-1), then forms
// S(j) and its softmax.  With both groups in the same order, the two waves of a SIMD meet in the same phase after every barrier (both in
// the score MFMAs, then both in the exponentials, then both in the PV MFMAs); rotated, a wave's vector phase lies beside its partner's MFMA
// phase in two of the three thirds of an interval.  V(j-1) has to outlive interval j, so the ring has THREE {K, V, subtree_end} slots
// (99 KB) and the loop is unrolled over them.
// =================================================================================================
constexpr int FWD4_LDS = 3 * (2 * TILE_BYTES + SE_BYTES);
template <int DT>
__global__ __launch_bounds__(512, 2) void tree_attn_fwd4_kernel(AttnParams p) {
  using T = Ty<DT>; using e = typename T::e; using v8 = typename T::v8; using v4 = typename T::v4;
  constexpr int NW = 8, BUF = 2 * TILE_BYTES + SE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[FWD4_LDS];

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hb = wave >> 2, rw = wave & 3;
  const int bid = blockIdx.x;
  const int hgroups = p.hgroups;
  const int kvh = bid % p.Hkv; const int rest = bid / p.Hkv; const int hgb = rest % hgroups;
  const int nqt = (p.Tq + DTA_QTILE - 1) / DTA_QTILE;
  const int qt = nqt - 1 - rest / hgroups;
  const int hq = kvh * p.group + p.head0 + hgb * 2 + hb;
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
  v8 pk[4];                                   // group 1: P of the previous tile, packed
  bool have_prev = false;

  int ck0 = it.k0, ckend = it.kend; bool cmask = it.masked();
  DTA_KV_DMA(smem, it.k0, NW)
  bool has_next = it.advance();
  // The Q fragments must be COMPLETE in hipcc's own book-keeping before the loop: it cannot see the asm DMA, but it does count the plain
  // global loads of Q, and with them still "pending" at the loop header it put `s_waitcnt vmcnt(7) .. vmcnt(0)` in front of the first eight
  // score MFMAs of the loop body - where vmcnt(0) also waits for the NEXT tile's DMA issued a few hundred cycles earlier.
#pragma unroll
  for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(qf[s]));
  DMA_WAIT(); __syncthreads();

#define FWD4_SCORES(KS)                                                                                    \
    const char* Ks = (KS);                                                                                 \
    const int* se_s = reinterpret_cast<const int*>(Ks + 2 * TILE_BYTES);                                   \
    f32x16 X[2];                                                                                           \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                     \
      _Pragma("unroll") for (int g = 0; g < 16; ++g) X[kb][g] = 0.f;                                       \
      _Pragma("unroll") for (int s = 0; s < 8; ++s)                                                        \
        X[kb] = T::mma(*reinterpret_cast<const v8*>(Ks + 8192 * kb + offs.row[s]), qf[s], X[kb]);         \
    }                                                                                                      \
    if (cmask) {                                                                                           \
      const int qlim = qidx < ckend ? qidx : ckend - 1;                                                    \
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
    float mx = max3(X[0][0], X[0][1], X[0][2]), mx1 = max3(X[1][0], X[1][1], X[1][2]);   /* two independent chains */ \
    _Pragma("unroll") for (int g = 3; g < 15; g += 2) { mx = max3(mx, X[0][g], X[0][g + 1]); mx1 = max3(mx1, X[1][g], X[1][g + 1]); } \
    mx = max3(mx, X[0][15], mx1);                                                                          \
    mx = half_max(fmaxf(mx, X[1][15]));        /* v_permlane32_swap: no LDS round trip (ds_bpermute + 6 address instructions before) */ \
    const float mc = mx * c;                                                                               \
    if (__builtin_expect(__any(mc > m + FWD_THR), 0)) {                                                    \
      const float mnew = fmaxf(m, mc);                                                                     \
      const float alpha = fast_exp2(m - mnew);                                                             \
      m = mnew; lsum *= alpha;                                                                             \
      _Pragma("unroll") for (int db = 0; db < 4; ++db)                                                     \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) O[db][g] *= alpha;                                  \
    }                                                                                                      \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                       \
      _Pragma("unroll") for (int g = 0; g < 16; ++g) { const float pv = fast_exp2(__builtin_fmaf(X[kb][g], c, -m)); lsum += pv; X[kb][g] = pv; }
#define FWD4_PV(VS, PB)                                                                                    \
    { const char* vs_ = (VS);                                                                              \
    _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4)                                                       \
      _Pragma("unroll") for (int db = 0; db < 4; ++db) O[db] = T::mma(tr_frag_o<v8>(vs_ + 4096 * s4, offs, db), PB[s4], O[db]); }
  // single-exit loops over a run-time ring slot (three compile-time slots x early exits made hipcc move the accumulators through scratch)
  int slot = 0;                               // wave-uniform
  bool more;
  if (hb == 0) {
    do {
      const int nslot = slot == 2 ? 0 : slot + 1;
      int nk0_ = 0, nkend_ = 0; bool nmask_ = false;
      if (has_next) { nk0_ = it.k0; nkend_ = it.kend; nmask_ = it.masked(); DTA_KV_DMA(smem + nslot * BUF, it.k0, NW) }
      FWD4_SCORES(smem + slot * BUF)
      v8 pb_[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) pb_[s4] = pack_half<DT>(X[s4 >> 1], s4 & 1);
      FWD4_PV(smem + slot * BUF + TILE_BYTES, pb_)
      DMA_WAIT(); __syncthreads();
      more = has_next;
      if (more) { ck0 = nk0_; ckend = nkend_; cmask = nmask_; has_next = it.advance(); slot = nslot; }
    } while (more);
  } else {
    int pslot = 0;
    do {
      const int nslot = slot == 2 ? 0 : slot + 1;
      int nk0_ = 0, nkend_ = 0; bool nmask_ = false;
      if (has_next) { nk0_ = it.k0; nkend_ = it.kend; nmask_ = it.masked(); DTA_KV_DMA(smem + nslot * BUF, it.k0, NW) }
      if (have_prev) FWD4_PV(smem + pslot * BUF + TILE_BYTES, pk)
      FWD4_SCORES(smem + slot * BUF)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) pk[s4] = pack_half<DT>(X[s4 >> 1], s4 & 1);
      have_prev = true;
      DMA_WAIT(); __syncthreads();
      more = has_next;
      pslot = slot;
      if (more) { ck0 = nk0_; ckend = nkend_; cmask = nmask_; has_next = it.advance(); slot = nslot; }
    } while (more);
    FWD4_PV(smem + pslot * BUF + TILE_BYTES, pk)      // the deferred product of the last tile (its slot is not written again)
  }
#undef FWD4_PV
#undef FWD4_SCORES

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
    if (h == 0) p.lse_w[(int64_t)hq * p.Tq + qrow] = m + __builtin_amdgcn_logf(lsum);
  }
}

// =================================================================================================
// forward, ONE wave per SIMD: a workgroup = 4 waves = 128 query rows, and every wave carries BOTH query heads of the kv group for its 32
// rows (512 registers per lane: O 2 x 64, Q fragments 2 x 32, scores 2 x 32).
//
// Why: in-kernel stamps of the 8-wave form above show that vector instructions do not run in the shadow of the SIMD
// PARTNER's MFMAs - a tile costs the SIMD its MFMA cycles PLUS both waves' vector-issue cycles - while they do run in the shadow of the
// wave's OWN MFMAs.  Here the two heads are two independent instruction chains inside one wave: every K and V fragment read feeds two
// MFMAs (half the LDS traffic per MFMA), and the exponentials / packs of both heads sit in the same basic block as the 32 PV MFMAs.
// Same tile iteration, masks, staging (LDS-DMA, double buffer, one barrier per tile) and outputs as the 8-wave form.
// =================================================================================================
// Score MFMAs with a VGPR destination, from inline asm: with a 512-register budget hipcc puts the accumulators of EVERY builtin MFMA into
// accumulation registers - right for O (128 registers nothing but the rare rescale touches), wrong for the scores, which the softmax
// reads element by element (8 v_accvgpr moves per MFMA in the loop).  hipcc does not know these are MFMAs: the wait states between the
// last one and the first vector read of its result (8-pass MFMA -> VALU: 11) are ours to provide - mfma_scores_done().
template <int DT> struct MmaV;
template <> struct MmaV<DTA_BF16> {
  static __device__ __forceinline__ void first(f32x16& d, bf16x8 a, bf16x8 b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b)); }
  static __device__ __forceinline__ void acc(f32x16& d, bf16x8 a, bf16x8 b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b)); }
};
template <> struct MmaV<DTA_F16> {
  static __device__ __forceinline__ void first(f32x16& d, f16x8 a, f16x8 b) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b)); }
  static __device__ __forceinline__ void acc(f32x16& d, f16x8 a, f16x8 b) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b)); }
};
__device__ __forceinline__ void mfma_scores_done(f32x16& a, f32x16& b, f32x16& c_, f32x16& d) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c_), "+v"(d));
}
constexpr float FWD3_THR = FWD_THR;          // the reference maximum follows a tile's row maximum only when that grew by more than this (log2 domain: P <= 16): the rescale of the 128 O accumulators (AGPR <-> VGPR moves) becomes rare

template <int DT>
__global__ __launch_bounds__(256, 1) void tree_attn_fwd3_kernel(AttnParams p) {
  using T = Ty<DT>; using e = typename T::e; using v8 = typename T::v8; using v4 = typename T::v4;
  constexpr int NW = 4, BUF = 2 * TILE_BYTES + SE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[QK_LDS];

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x;
  const int hgroups = p.hgroups;
  const int kvh = bid % p.Hkv; const int rest = bid / p.Hkv; const int hgb = rest % hgroups;
  const int nqt = (p.Tq + DTA_QTILE - 1) / DTA_QTILE;
  const int qt = nqt - 1 - rest / hgroups;
  const int hq0 = kvh * p.group + p.head0 + hgb * 2;                 // this workgroup's two query heads: hq0, hq0 + 1
  const int q0 = qt * DTA_QTILE;
  const int qrow = q0 + wave * 32 + r;
  const int qrow_c = qrow < p.Tq ? qrow : p.Tq - 1;
  const int qidx = p.q_offset + qrow;

  TileIter it; it.runs = p.runs; it.diag_first_q = p.q_offset + q0;
  if (p.runs) { it.ri = p.run_ptr[qt]; it.re = p.run_ptr[qt + 1]; if (!it.load_run()) return; }
  else { it.ri = 0; it.re = 1; it.k0 = 0; it.flag = 1; int last = p.q_offset + (q0 + DTA_QTILE < p.Tq ? q0 + DTA_QTILE : p.Tq); it.kend = last < p.Tk ? last : p.Tk; if (it.kend <= 0) return; }

  v8 qf[2][8];
#pragma unroll
  for (int hd = 0; hd < 2; ++hd) {
    const e* qp = reinterpret_cast<const e*>(p.q) + (int64_t)qrow_c * p.q_st + (int64_t)(hq0 + hd) * p.q_sh;
#pragma unroll
    for (int s = 0; s < 8; ++s) { qf[hd][s] = *reinterpret_cast<const v8*>(qp + 16 * s + 8 * h); asm volatile("" : "+a"(qf[hd][s])); }   // Q fragments live in accumulation registers (pure MFMA operands)
  }
  const e* kbase = reinterpret_cast<const e*>(p.k) + (int64_t)kvh * p.kv_sh;
  const e* vbase = reinterpret_cast<const e*>(p.v) + (int64_t)kvh * p.v_sh;
  const FragOffs offs = frag_offsets(lane);
  DTA_KV_OFFSETS(NW)

  f32x16 O[2][4];
#pragma unroll
  for (int hd = 0; hd < 2; ++hd)
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 16; ++g) O[hd][db][g] = 0.f;
  float m[2] = {-1e30f, -1e30f}, lsum[2] = {0.f, 0.f};
  const float c = p.scale * LOG2E;

  int ck0 = it.k0, ckend = it.kend; bool cmask = it.masked();
  DTA_KV_DMA(smem, it.k0, NW)
  bool has_next = it.advance();
  DMA_WAIT(); __syncthreads();

  int cur = 0;
  bool more = true;
  do {          // ONE straight-line body with a single exit at the bottom and a run-time buffer toggle: with two unrolled bodies and exits from the
                // middle hipcc carried the 128 O accumulators in VGPRs between them (128 v_accvgpr moves each way per tile)
    int nk0_ = 0, nkend_ = 0; bool nmask_ = false;
    if (has_next) { nk0_ = it.k0; nkend_ = it.kend; nmask_ = it.masked(); }
    DTA_KV_DMA_PREP(smem + (cur ^ 1) * BUF, has_next ? it.k0 : 0, NW, has_next)
    const char* Ks = smem + cur * BUF; const char* Vs = Ks + TILE_BYTES;
    const int* se_s = reinterpret_cast<const int*>(Ks + 2 * TILE_BYTES);
    f32x16 X[2][2];
    { v8 kf[2][8];                           /* all 16 K fragments requested up front (64 registers): one LDS latency per tile, not one per MFMA pair */
      #pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        #pragma unroll
      for (int s = 0; s < 8; ++s) kf[kb][s] = *reinterpret_cast<const v8*>(Ks + 8192 * kb + offs.row[s]);
      #pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        #pragma unroll
      for (int s = 0; s < 8; ++s) {                    /* one K fragment, two MFMAs */
          if (s == 0) { MmaV<DT>::first(X[0][kb], kf[kb][s], qf[0][s]); MmaV<DT>::first(X[1][kb], kf[kb][s], qf[1][s]); }
          else { MmaV<DT>::acc(X[0][kb], kf[kb][s], qf[0][s]); MmaV<DT>::acc(X[1][kb], kf[kb][s], qf[1][s]); }
          if (s == 3 && has_next) { if (kb == 0) { DTA_KV_DMA_PAIR(0, NW) } else { DTA_KV_DMA_PAIR(1, NW) } }      // the next tile's DMA pieces go out between the MFMAs
        }
    }
    mfma_scores_done(X[0][0], X[0][1], X[1][0], X[1][1]);
    if (cmask) {
      const int qlim = qidx < ckend ? qidx : ckend - 1;
      #pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        #pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
          const int kl = 32 * kb + 8 * gq + 4 * h;
          const int4 se4 = *reinterpret_cast<const int4*>(se_s + kl);
          const int sev[4] = {se4.x, se4.y, se4.z, se4.w};
          #pragma unroll
      for (int j = 0; j < 4; ++j) {
            const bool ok = (ck0 + kl + j <= qlim) && (qidx < sev[j]);
            X[0][kb][4 * gq + j] = ok ? X[0][kb][4 * gq + j] : -INFINITY;
            X[1][kb][4 * gq + j] = ok ? X[1][kb][4 * gq + j] : -INFINITY;
          }
        }
    }
    float mc[2];
    #pragma unroll
      for (int hd = 0; hd < 2; ++hd) {
      float mx = max3(X[hd][0][0], X[hd][0][1], X[hd][0][2]);
      #pragma unroll
      for (int g = 3; g < 15; g += 2) mx = max3(mx, X[hd][0][g], X[hd][0][g + 1]);
      mx = fmaxf(mx, X[hd][0][15]);
      #pragma unroll
      for (int g = 0; g < 16; g += 2) mx = max3(mx, X[hd][1][g], X[hd][1][g + 1]);
      mc[hd] = half_max(mx) * c;
    }
    if (__builtin_expect(__any((mc[0] > m[0] + FWD3_THR) || (mc[1] > m[1] + FWD3_THR)), 0)) {   /* COLD side path, both heads at once: the O <-> VGPR moves belong in here */
      #pragma unroll
      for (int hd = 0; hd < 2; ++hd) {
        const float mnew = fmaxf(m[hd], mc[hd]);
        const float alpha = fast_exp2(m[hd] - mnew);
        m[hd] = mnew; lsum[hd] *= alpha;
        #pragma unroll
      for (int db = 0; db < 4; ++db)
          #pragma unroll
      for (int g = 0; g < 16; ++g) O[hd][db][g] *= alpha;
      }
    }
    /* exponentials / packs of both heads and the 32 PV MFMAs: one basic block, every V fragment read feeds two MFMAs */
    #pragma unroll
      for (int hd = 0; hd < 2; ++hd)
      #pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        #pragma unroll
      for (int g = 0; g < 16; ++g) { const float pv = fast_exp2(__builtin_fmaf(X[hd][kb][g], c, -m[hd])); lsum[hd] += pv; X[hd][kb][g] = pv; }
    #pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
      const v8 pb0 = pack_half<DT>(X[0][s4 >> 1], s4 & 1);
      const v8 pb1 = pack_half<DT>(X[1][s4 >> 1], s4 & 1);
      #pragma unroll
      for (int db = 0; db < 4; ++db) {
        const v8 vf = tr_frag_o<v8>(Vs + 4096 * s4, offs, db);
        O[0][db] = T::mma(vf, pb0, O[0][db]);
        O[1][db] = T::mma(vf, pb1, O[1][db]);
      }
    }
    /* the loop-carried home of O is the accumulation file (hipcc otherwise carries it in VGPRs and moves 128 registers each way per tile) */
    #pragma unroll
      for (int db = 0; db < 4; ++db) { asm volatile("" : "+a"(O[0][db])); asm volatile("" : "+a"(O[1][db])); }
    DMA_WAIT(); __syncthreads();
    more = has_next;
    ck0 = nk0_; ckend = nkend_; cmask = nmask_; cur ^= 1;
    if (more) has_next = it.advance();
  } while (more);

#pragma unroll
  for (int hd = 0; hd < 2; ++hd) {
    float l = lsum[hd];
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    if (qrow < p.Tq) {
      e* op = reinterpret_cast<e*>(p.out) + (int64_t)qrow * p.o_st + (int64_t)(hq0 + hd) * p.o_sh;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          v4 w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = (e)(O[hd][db][4 * gq + j] * inv);
          *reinterpret_cast<v4*>(op + 32 * db + 8 * gq + 4 * h) = w;
        }
      if (h == 0) p.lse_w[(int64_t)(hq0 + hd) * p.Tq + qrow] = m[hd] + __builtin_amdgcn_logf(l);
    }
  }
}

// =================================================================================================
// backward part 1: delta + dQ   (query tile owns the workgroup; same sweep as the forward)
// =================================================================================================
template <int DT, int HPB>
__global__ __launch_bounds__(256 * HPB, 2) void tree_attn_bwd_dq_kernel(AttnParams p) {
  using T = Ty<DT>; using e = typename T::e; using v8 = typename T::v8; using v4 = typename T::v4;
  __shared__ __attribute__((aligned(16))) char smem[QK_LDS];

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform values live in SGPRs
  DTA_PRIO_YOUNGER_HALF(wave)
  const int hb = wave >> 2, rw = wave & 3;
  const int bid = blockIdx.x;
  const int hgroups = p.hgroups;
  const int kvh = bid % p.Hkv; const int rest = bid / p.Hkv; const int hgb = rest % hgroups;
  const int nqt = (p.Tq + DTA_QTILE - 1) / DTA_QTILE;
  const int qt = nqt - 1 - rest / hgroups;
  const int hq = kvh * p.group + p.head0 + hgb * HPB + hb;
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
  if (h == 0 && qrow < p.Tq) p.delta[(int64_t)hq * p.Tq + qrow] = -delta;    // workspace holds -delta: the dK/dV kernel loads it as the INITIAL dP accumulator

  TileIter it; it.runs = p.runs; it.diag_first_q = p.q_offset + q0;
  bool any = true;
  if (p.runs) { it.ri = p.run_ptr[qt]; it.re = p.run_ptr[qt + 1]; any = it.load_run(); }
  else { it.ri = 0; it.re = 1; it.k0 = 0; it.flag = 1; int last = p.q_offset + (q0 + DTA_QTILE < p.Tq ? q0 + DTA_QTILE : p.Tq); it.kend = last < p.Tk ? last : p.Tk; any = it.kend > 0; }

  const e* kbase = reinterpret_cast<const e*>(p.k) + (int64_t)kvh * p.kv_sh;
  const e* vbase = reinterpret_cast<const e*>(p.v) + (int64_t)kvh * p.v_sh;
  constexpr int NW = DTA_DMA_WAVES(HPB), BUF = 2 * TILE_BYTES + SE_BYTES;
  DTA_KV_OFFSETS(NW)                      // K/V tiles by LDS-DMA as in the forward (no staging registers, no ds_write)

  f32x16 DQ[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int g = 0; g < 16; ++g) DQ[db][g] = 0.f;
  const float c = p.scale * LOG2E;

  // dP starts at -delta (this lane's query row) instead of 0: dS/scale = p * dP' costs one multiply per element, and the softmax
  // scale goes onto dQ once, in the epilogue (as the dK/dV kernel does for dK)
  f32x16 DI;
#pragma unroll
  for (int g = 0; g < 16; ++g) DI[g] = -delta;
  if (any) {
    int ck0 = it.k0, ckend = it.kend; bool cmask = it.masked();
    DTA_KV_DMA(smem, it.k0, NW)
    bool has_next = it.advance();
    DMA_WAIT(); __syncthreads();
    int cur = 0;
    while (true) {
      int nk0 = 0, nkend = 0; bool nmask = false;
      if (has_next) { nk0 = it.k0; nkend = it.kend; nmask = it.masked(); DTA_KV_DMA(smem + (cur ^ 1) * BUF, it.k0, NW) }
      const char* Ks = smem + cur * BUF; const char* Vs = Ks + TILE_BYTES;
      const int* se_s = reinterpret_cast<const int*>(Ks + 2 * TILE_BYTES);
      // one 32-key block at a time keeps S^T/dP^T at 32 live accumulators (2 waves per SIMD need <= 256 registers)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16 X, DP;
#pragma unroll
        for (int g = 0; g < 16; ++g) X[g] = 0.f;
        {                                                  // fragment reads one k-step ahead of the MFMAs that use them
          v8 kf_ = row_frag<v8>(Ks, 32 * kb + r, h), vf_ = row_frag<v8>(Vs, 32 * kb + r, h);
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            v8 nk_ = kf_, nv_ = vf_;
            if (s < 7) { nk_ = row_frag<v8>(Ks, 32 * kb + r, 2 * s + 2 + h); nv_ = row_frag<v8>(Vs, 32 * kb + r, 2 * s + 2 + h); }
            X = T::mma(kf_, qf[s], X);
            DP = T::mma(vf_, dof[s], s == 0 ? DI : DP);
            kf_ = nk_; vf_ = nv_;
          }
          // hipcc's scheduler otherwise sinks every read pair directly in front of its two MFMAs (one register pair re-used: each MFMA
          // pair then waits a full LDS latency): pin the order {4 reads} {2 MFMA, 2 reads} x 6 {4 MFMA}
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
          for (int i_ = 0; i_ < 6; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        // dS^T / scale = P ∘ (dP − delta); the interval mask only on tiles of runs flagged partial
        if (cmask) {
          const int qlim = qidx < ckend ? qidx : ckend - 1;      // keys at or beyond the run end never count
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int kl = 32 * kb + 8 * gq + 4 * h;
            const int4 se4 = *reinterpret_cast<const int4*>(se_s + kl);
            const int sev[4] = {se4.x, se4.y, se4.z, se4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int g = 4 * gq + j;
              const bool ok = (ck0 + kl + j <= qlim) && (qidx < sev[j]);
              const float pv = ok ? fast_exp2(__builtin_fmaf(X[g], c, -lse2)) : 0.f;
              X[g] = pv * DP[g];
            }
          }
        } else {
#pragma unroll
          for (int g = 0; g < 16; ++g) X[g] = fast_exp2(__builtin_fmaf(X[g], c, -lse2)) * DP[g];
        }
        // dQ^T[d][q] += K^T · dS^T
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const v8 db_ = pack_half<DT>(X, s2);
#pragma unroll
          for (int db = 0; db < 4; ++db) DQ[db] = T::mma(tr_frag<v8>(Ks, 32 * kb + 16 * s2, db, lane), db_, DQ[db]);
        }
      }
      DMA_WAIT(); __syncthreads();
      if (!has_next) break;
      cur ^= 1; ck0 = nk0; ckend = nkend; cmask = nmask;
      has_next = it.advance();
    }
  }
  if (qrow < p.Tq) {
    e* dqp = reinterpret_cast<e*>(p.dq) + (int64_t)qrow * p.dq_st + (int64_t)hq * p.dq_sh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        v4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (e)(DQ[db][4 * gq + j] * p.scale);
        *reinterpret_cast<v4*>(dqp + 32 * db + 8 * gq + 4 * h) = w;
      }
  }
}

// =================================================================================================
// backward part 2: dK, dV.  Key tile owns the workgroup.
// =================================================================================================
// -------------------------------------------------------------------------------------------------
// dK/dV with TWO waves per SIMD (8 waves): the two wave groups own the same 128 keys and split every 64-row query tile
// between them (group g takes rows 32g..32g+31), so they share ONE double-buffered Q/dO image.  To fit 256 registers
// the K/V fragments (pure MFMA B operands) live in LDS in fragment order (one lane-linear, conflict-free ds_read_b128
// per use) instead of 64 registers.  The groups' partial dK/dV are summed through LDS in a fixed order at the end.
// -------------------------------------------------------------------------------------------------
// dkv2's S / dP chains: left alone (-DDTA_KV2_PIN=0) hipcc runs most of the dP MFMAs and then the S MFMAs as two dependent chains, every operand
// read directly in front of the MFMA that uses it.  Default 2: reads stay just in time but the two chains ALTERNATE (-1.0 %, three alternating
// pairs on one box: 1.3407 -> 1.3272 ms); 1: {delta row + operands of two k-steps} {2 MFMA, 4 reads} x 6 {4 MFMA} - read bursts, 5 spills, 7 % slower.
#ifndef DTA_KV2_PIN
#define DTA_KV2_PIN 2
#endif
#if DTA_KV2_PIN == 2      /* just-in-time reads, but the two chains alternating: {delta row} {2 reads, 1 MFMA} x 16 */
#define DTA_KV2_PIN_ORDER                                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
#elif DTA_KV2_PIN
#define DTA_KV2_PIN_ORDER                                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);                                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); } \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#else
#define DTA_KV2_PIN_ORDER
#endif

constexpr int KV2_FRAGS = 4 * 16384;                                // 4 key slots x {K: 8 fragments x 1 KiB, V: 8 x 1 KiB}
constexpr int KV2_BUF = 2 * TILE_BYTES + 512;
constexpr int KV2_LDS = KV2_FRAGS + 2 * KV2_BUF + 16;                   // + se_min[4]: ONE __shared__ object (a second one makes hipcc drain vmcnt in front of every LDS read)

template <int DT>
__global__ __launch_bounds__(512, 2) void tree_attn_bwd_dkv2_kernel(AttnParams p) {
  using T = Ty<DT>; using e = typename T::e; using v8 = typename T::v8; using v4 = typename T::v4;
  constexpr int KT = 128;
  __shared__ __attribute__((aligned(16))) char smem_all[KV2_LDS];
  const int tid8 = threadIdx.x, tid = tid8 & 255, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid8 >> 6);       // 0..7
  DTA_PRIO_YOUNGER_HALF(wave8)
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
  int* se_min_s = reinterpret_cast<int*>(smem_all + KV2_FRAGS + 2 * KV2_BUF);
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

  // tile DMA: 16 one-KiB pieces per image over 8 waves = 2 per wave per image (piece = 2*wave8 + i: rows 8*wave8 + 4*i ..);
  // lse / delta rows (64 floats each) by 4-byte DMA from waves 0 / 1.
  uint32_t voff_q[2], voff_d[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row_ = 8 * wave8 + 4 * i + (lane >> 4);
    voff_q[i] = dma_src_off(row_, row_, lane, p.q_st, sizeof(e));
    voff_d[i] = dma_src_off(row_, row_, lane, p.o_st, sizeof(e));
  }
  const uint32_t lds_tiles = lds_addr(smem);
  // scalar cursor of the NEXT tile to stage: byte offsets of its first row in q / dout and float offset of its row
  // constants; advanced by additions (64 rows down, or to row qbeg of the next query head) - no 64-bit multiplies per tile
  const int64_t q_step = 64 * p.q_st * (int64_t)sizeof(e), d_step = 64 * p.o_st * (int64_t)sizeof(e);
  const int64_t q_wrap = p.q_sh * (int64_t)sizeof(e) - ntile * q_step, d_wrap = p.o_sh * (int64_t)sizeof(e) - ntile * d_step;
  int64_t q_cur = ((int64_t)(kvh * p.group) * p.q_sh + (int64_t)(qbeg - p.q_offset) * p.q_st) * (int64_t)sizeof(e);
  int64_t d_cur = ((int64_t)(kvh * p.group) * p.o_sh + (int64_t)(qbeg - p.q_offset) * p.o_st) * (int64_t)sizeof(e);
  int64_t c_cur = (int64_t)(kvh * p.group) * p.Tq;
  int row_n = qbeg - p.q_offset, ti_n = 0;
#define KV2_DMA(B)                                                                                         \
  { const uint32_t lb_ = lds_tiles + (uint32_t)(B) * KV2_BUF;                                              \
    if (wave8 < 2) { int qr_ = row_n + lane; qr_ = qr_ < p.Tq ? qr_ : p.Tq - 1;   /* wave 0: lse[64], wave 1: delta[64] */ \
      dma_dword((uint32_t)qr_ * 4u, (wave8 == 0 ? p.lse_r : p.delta) + c_cur, lb_ + 2 * TILE_BYTES + wave8 * 256); } \
    uint32_t oq0_ = voff_q[0], oq1_ = voff_q[1], od0_ = voff_d[0], od1_ = voff_d[1];                       \
    if (row_n + 64 > p.Tq) {                       /* ragged last tile of the tensor: clamp the row per lane */ \
      const int ra_ = 8 * wave8 + (lane >> 4), rb_ = ra_ + 4;                                              \
      const int ca_ = row_n + ra_ < p.Tq ? ra_ : p.Tq - 1 - row_n, cb_ = row_n + rb_ < p.Tq ? rb_ : p.Tq - 1 - row_n; \
      oq0_ = dma_src_off(ca_, ra_, lane, p.q_st, sizeof(e)); oq1_ = dma_src_off(cb_, rb_, lane, p.q_st, sizeof(e)); \
      od0_ = dma_src_off(ca_, ra_, lane, p.o_st, sizeof(e)); od1_ = dma_src_off(cb_, rb_, lane, p.o_st, sizeof(e)); } \
    dma_pair2(oq0_, oq1_, reinterpret_cast<const char*>(p.q) + q_cur, od0_, od1_, reinterpret_cast<const char*>(p.dout) + d_cur, lb_ + wave8 * 2048); \
    ++ti_n; row_n += 64; q_cur += q_step; d_cur += d_step;                                                 \
    if (ti_n >= ntile) { ti_n = 0; row_n = qbeg - p.q_offset; q_cur += q_wrap; d_cur += d_wrap; c_cur += p.Tq; } }

  // Per-lane LDS byte offsets of every fragment read of this wave group inside a tile buffer, computed once; the tile loop is
  // unrolled over the two buffers so that the buffer offset is an instruction immediate (no per-tile address VALU).
  int ar[8], at[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ar[j] = offs.row[j] + grp * (32 * 256); at[j] = offs.tr[j] + grp * (32 * 256); }
  const int rc_off = (32 * grp + 4 * h) * 4;                            // this lane's first row constant (lse / -delta) inside a buffer

  // one 64-row query tile out of buffer BUFI (compile-time)
#define KV2_TILE(BUFI)                                                                                     \
  {                                                                                                        \
    const int ti = ti_c;                                                                                   \
    ti_c += 1;                                                                                             \
    if (ti_c >= ntile) ti_c = 0;                                                                           \
    if (idx + 1 < total) KV2_DMA(1 - (BUFI))      /* lands while this tile computes; waited for at the tile end */ \
    const char* tb = smem + (BUFI) * KV2_BUF;                                                              \
    const char* rc = tb + 2 * TILE_BYTES + rc_off;                                                         \
    const int qi0 = qbeg + 64 * ti + 32 * grp;                       /* packed index of this group's first row */ \
    const bool full = (qbeg + 64 * ti >= k0 + KT - 1) && (qbeg + 64 * ti + 63 < se_min);   /* workgroup-uniform: no mask needed */ \
    /* S starts at 0 (inline constant); dP starts at -delta, read from LDS straight into the accumulator registers:    \
       p = exp2(c*S - lse),  dS/scale = p * dP'  with dP' = dO.V^T - delta  (the softmax scale of dS goes onto dK once, in the epilogue) */ \
    f32x16 S, DP;                                                                                          \
    _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                     \
      const float4 d4 = *reinterpret_cast<const float4*>(rc + 256 + 32 * gq);                              \
      DP[4 * gq] = d4.x; DP[4 * gq + 1] = d4.y; DP[4 * gq + 2] = d4.z; DP[4 * gq + 3] = d4.w;              \
    }                                                                                                      \
    _Pragma("unroll") for (int g = 0; g < 16; ++g) S[g] = 0.f;                                             \
    _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                                        \
      const v8 aq = *reinterpret_cast<const v8*>(tb + ar[s]);                                              \
      const v8 ad = *reinterpret_cast<const v8*>(tb + ar[s] + TILE_BYTES);                                 \
      const v8 kfs = *reinterpret_cast<const v8*>(kvs + s * 1024 + lane * 16);                             \
      const v8 vfs = *reinterpret_cast<const v8*>(kvs + 8192 + s * 1024 + lane * 16);                      \
      S = T::mma(aq, kfs, S); DP = T::mma(ad, vfs, DP);                                                    \
    }                                                                                                      \
    DTA_KV2_PIN_ORDER                                                                                      \
    float nl[16];                                                                                          \
    _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                     \
      const float4 l4 = *reinterpret_cast<const float4*>(rc + 32 * gq);                                    \
      nl[4 * gq] = l4.x; nl[4 * gq + 1] = l4.y; nl[4 * gq + 2] = l4.z; nl[4 * gq + 3] = l4.w;              \
    }                                                                                                      \
    if (full) {                                                                                            \
      _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                     \
        const float pv = fast_exp2(__builtin_fmaf(S[g], c, -nl[g]));                                       \
        S[g] = pv;                                                                                         \
        DP[g] = pv * DP[g];                                                                                \
      }                                                                                                    \
    } else {                                                                                               \
      _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                     \
        const int qi = qi0 + 8 * (g >> 2) + 4 * h + (g & 3);                                               \
        const bool ok = (kidx <= qi) && (qi < se_l);                                                       \
        const float pv = ok ? fast_exp2(__builtin_fmaf(S[g], c, -nl[g])) : 0.f;                            \
        S[g] = pv;                                                                                         \
        DP[g] = pv * DP[g];                                                                                \
      }                                                                                                    \
    }                                                                                                      \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                     \
      const v8 pb = pack_half<DT>(S, s2), sbf = pack_half<DT>(DP, s2);                                     \
      _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                                   \
        const v8 adt = tr_pair<v8>(tb + at[db] + TILE_BYTES + 4096 * s2, tb + at[4 + db] + TILE_BYTES + 4096 * s2); \
        const v8 aqt = tr_pair<v8>(tb + at[db] + 4096 * s2, tb + at[4 + db] + 4096 * s2);                  \
        DV[db] = T::mma(adt, pb, DV[db]); DK[db] = T::mma(aqt, sbf, DK[db]);                               \
      }                                                                                                    \
    }                                                                                                      \
    DMA_WAIT(); __syncthreads();                                                                           \
    ++idx;                                                                                                 \
  }

  {
    int ti_c = 0;
    if (total > 0) KV2_DMA(0)
    DMA_WAIT(); __syncthreads();
    int idx = 0;
    while (idx < total) {
      KV2_TILE(0)
      if (idx >= total) break;
      KV2_TILE(1)
    }
  }
#undef KV2_TILE
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
  } else if (kidx < p.Tk && p.accumulate == 2) {
    // fp32 accumulation buffers (the grad-KV stack of the block-wise engine: hundreds of adds per row stay exact to fp32)
    float* dkp = reinterpret_cast<float*>(p.dk) + (int64_t)kidx * p.dkv_st + (int64_t)kvh * p.dkv_sh;
    float* dvp = reinterpret_cast<float*>(p.dv) + (int64_t)kidx * p.dkv_st + (int64_t)kvh * p.dkv_sh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d = 32 * db + 8 * gq + 4 * h;
        float4 a = *reinterpret_cast<const float4*>(dkp + d), b = *reinterpret_cast<const float4*>(dvp + d);
        a.x += DK[db][4 * gq]; a.y += DK[db][4 * gq + 1]; a.z += DK[db][4 * gq + 2]; a.w += DK[db][4 * gq + 3];
        b.x += DV[db][4 * gq]; b.y += DV[db][4 * gq + 1]; b.z += DV[db][4 * gq + 2]; b.w += DV[db][4 * gq + 3];
        *reinterpret_cast<float4*>(dkp + d) = a;
        *reinterpret_cast<float4*>(dvp + d) = b;
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
    if (p.accumulate == 2) {
      float* outf = reinterpret_cast<float*>(which ? p.dv : p.dk) + (int64_t)kidx * p.dkv_st + (int64_t)kvh * p.dkv_sh + d;
      float4 o = *reinterpret_cast<const float4*>(outf);
      o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
      *reinterpret_cast<float4*>(outf) = o;
      continue;
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
  if (head_dim != 128 || Hq % Hkv != 0 || (dtype != DTA_BF16 && dtype != DTA_F16 && dtype != DTA_F32)) return DTA_EUNSUPPORTED;
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out) || (q_st | q_sh | kv_st | kv_sh | v_st | v_sh | o_st | o_sh) % 8 != 0) return DTA_EALIGN;
  if (dtype == DTA_F32) {                                    // fp32 models: the plain-FMA correctness path (tree_attn_f32.hip)
    DTA_REFUSE_IF_PRIOR_ERROR();
    return dta_attn_fwd_f32(q, k, v, out, lse, subtree_end, run_ptr, runs, Tq, Tk, q_offset, Hq, Hkv, q_st, q_sh, kv_st, kv_sh, v_st, v_sh, o_st, o_sh,
                            scale, static_cast<hipStream_t>(stream));
  }
  // the tile DMA addresses a 64-row tile as scalar base + 32-bit lane offset: token strides must keep 64 rows inside 4 GiB
  if (kv_st < 0 || v_st < 0 || kv_st > (1 << 24) || v_st > (1 << 24)) return DTA_EUNSUPPORTED;
  AttnParams p{};
  p.q = q; p.k = k; p.v = v; p.out = out; p.lse_w = lse; p.subtree_end = subtree_end; p.run_ptr = run_ptr; p.runs = runs;
  p.Tq = Tq; p.Tk = Tk; p.q_offset = q_offset; p.Hq = Hq; p.Hkv = Hkv; p.group = Hq / Hkv;
  p.q_st = q_st; p.q_sh = q_sh; p.kv_st = kv_st; p.kv_sh = kv_sh; p.v_st = v_st; p.v_sh = v_sh; p.o_st = o_st; p.o_sh = o_sh; p.scale = scale;
  const int nqt = (Tq + DTA_QTILE - 1) / DTA_QTILE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DTA_REFUSE_IF_PRIOR_ERROR();
  // two query heads of a kv group share the staged K/V tiles (512 threads); an odd group sends its last head through the
  // one-head form in a second launch (Qwen3-14B: 40 query / 8 kv heads = 2 pairs + 1 per group)
  const int npair = p.group / 2;
  if (npair > 0) {
    p.hgroups = npair; p.head0 = 0;
    dim3 grid(nqt * Hkv * npair), block(512);
    static const int form = [] { const char* e_ = getenv("DTA_FWD_FORM"); return e_ ? atoi(e_) : DTA_FWD_FORM_DEFAULT; }();   // 1: 8 waves, one head each; 3: 4 waves, two heads each, one wave per SIMD; 4: 8 waves, head groups half a tile apart (A/B switch)
    if (form == 3) {
      if (dtype == DTA_BF16) hipLaunchKernelGGL((tree_attn_fwd3_kernel<DTA_BF16>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((tree_attn_fwd3_kernel<DTA_F16>), grid, dim3(256), 0, st, p);
    } else if (form == 4) {
      if (dtype == DTA_BF16) hipLaunchKernelGGL((tree_attn_fwd4_kernel<DTA_BF16>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((tree_attn_fwd4_kernel<DTA_F16>), grid, block, 0, st, p);
    } else {
      if (dtype == DTA_BF16) hipLaunchKernelGGL((tree_attn_fwd_kernel<DTA_BF16, 2>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((tree_attn_fwd_kernel<DTA_F16, 2>), grid, block, 0, st, p);
    }
  }
  if (p.group % 2) {
    p.hgroups = 1; p.head0 = p.group - 1;
    dim3 grid(nqt * Hkv), block(256);
    if (dtype == DTA_BF16) hipLaunchKernelGGL((tree_attn_fwd_kernel<DTA_BF16, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((tree_attn_fwd_kernel<DTA_F16, 1>), grid, block, 0, st, p);
  }
  return DTA_LAUNCH_STATUS();
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
  if (head_dim != 128 || Hq % Hkv != 0 || (dtype != DTA_BF16 && dtype != DTA_F16 && dtype != DTA_F32) || accumulate < 0 || accumulate > 2) return DTA_EUNSUPPORTED;
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out) || !aligned16(dout) || !aligned16(dq) || !aligned16(dk) || !aligned16(dv) ||
      (q_st | q_sh | kv_st | kv_sh | v_st | v_sh | o_st | o_sh | dq_st | dq_sh | dkv_st | dkv_sh) % 8 != 0) return DTA_EALIGN;
  if (dtype == DTA_F32) {
    if ((which & 7) == 0) return DTA_EINVAL;
    DTA_REFUSE_IF_PRIOR_ERROR();
    return dta_attn_bwd_f32(q, k, v, out, dout, lse, delta, dq, dk, dv, subtree_end, run_ptr, runs, ktile_qend, Tq, Tk, q_offset, Hq, Hkv,
                            q_st, q_sh, kv_st, kv_sh, v_st, v_sh, o_st, o_sh, dq_st, dq_sh, dkv_st, dkv_sh, scale, accumulate, which,
                            static_cast<hipStream_t>(stream));
  }
  if (q_st < 0 || o_st < 0 || q_st > (1 << 24) || o_st > (1 << 24)) return DTA_EUNSUPPORTED;   // 64-row tile = scalar base + 32-bit lane offset
  AttnParams p{};
  p.q = q; p.k = k; p.v = v; p.o = out; p.dout = dout; p.lse_r = lse; p.delta = delta; p.dq = dq; p.dk = dk; p.dv = dv;
  p.subtree_end = subtree_end; p.run_ptr = run_ptr; p.runs = runs; p.ktile_qend = ktile_qend;
  p.dkv_units = dkv_units; p.dkv_splits = dkv_splits; p.dkv_ws = dkv_ws;
  p.Tq = Tq; p.Tk = Tk; p.q_offset = q_offset; p.Hq = Hq; p.Hkv = Hkv; p.group = Hq / Hkv;
  p.q_st = q_st; p.q_sh = q_sh; p.kv_st = kv_st; p.kv_sh = kv_sh; p.v_st = v_st; p.v_sh = v_sh; p.o_st = o_st; p.o_sh = o_sh;
  p.dq_st = dq_st; p.dq_sh = dq_sh; p.dkv_st = dkv_st; p.dkv_sh = dkv_sh; p.scale = scale; p.accumulate = accumulate;
  const int nqt = (Tq + DTA_QTILE - 1) / DTA_QTILE;
  p.ktile = DTA_KTILE;
  const int nkt = (Tk + p.ktile - 1) / p.ktile;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DTA_REFUSE_IF_PRIOR_ERROR();
  if ((which & 7) == 0) return DTA_EINVAL;
  const bool fin = ((which & 2) && !(which & 8)) || (which & 4);     // slab finalize: with the dK/dV launch unless bit3, or alone (bit2)
  const int ndkv = dkv_units ? n_units : nkt;
  const int npair = p.group / 2;                                     // as in the forward: head pairs, then the odd head alone
  AttnParams pp = p, ps = p;
  pp.hgroups = npair; pp.head0 = 0; ps.hgroups = 1; ps.head0 = p.group - 1;
  const dim3 gqp(nqt * Hkv * (npair > 0 ? npair : 1)), gqs(nqt * Hkv);
  if (dtype == DTA_BF16) {
    if (which & 1) {
      if (npair > 0) hipLaunchKernelGGL((tree_attn_bwd_dq_kernel<DTA_BF16, 2>), gqp, dim3(512), 0, st, pp);
      if (p.group % 2) hipLaunchKernelGGL((tree_attn_bwd_dq_kernel<DTA_BF16, 1>), gqs, dim3(256), 0, st, ps);
    }
    if (which & 2) hipLaunchKernelGGL((tree_attn_bwd_dkv2_kernel<DTA_BF16>), dim3(ndkv * Hkv), dim3(512), 0, st, p);
    if (fin && dkv_units && n_splits > 0) hipLaunchKernelGGL(tree_attn_bwd_dkv_finalize_kernel<DTA_BF16>, dim3(n_splits * Hkv, FIN_SPLIT), dim3(256), 0, st, p);
  } else {
    if (which & 1) {
      if (npair > 0) hipLaunchKernelGGL((tree_attn_bwd_dq_kernel<DTA_F16, 2>), gqp, dim3(512), 0, st, pp);
      if (p.group % 2) hipLaunchKernelGGL((tree_attn_bwd_dq_kernel<DTA_F16, 1>), gqs, dim3(256), 0, st, ps);
    }
    if (which & 2) hipLaunchKernelGGL((tree_attn_bwd_dkv2_kernel<DTA_F16>), dim3(ndkv * Hkv), dim3(512), 0, st, p);
    if (fin && dkv_units && n_splits > 0) hipLaunchKernelGGL(tree_attn_bwd_dkv_finalize_kernel<DTA_F16>, dim3(n_splits * Hkv, FIN_SPLIT), dim3(256), 0, st, p);
  }
  return DTA_LAUNCH_STATUS();
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
