// Trie build kernels for gfx950: integer / index work, HBM-bound, bit-exact.
//   dta_lcp_adjacent   — first mismatch of adjacent sorted sequences by wave ballot
//   dta_leafize        — leafization as a ballot/prefix-scan stream compaction
//   dta_preorder_meta  — packed pre-order token / depth / parent / subtree_end gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dta_common.h"

namespace {

// One 256-thread workgroup per adjacent pair.  Every step compares 1024 positions (4 coalesced
// 8-byte loads per lane per sequence in flight), ballots the mismatches per wave and stops at the
// first step that holds one.
__global__ __launch_bounds__(256) void lcp_adjacent_kernel(const int64_t* __restrict__ tokens, const int64_t* __restrict__ starts,
                                                           const int32_t* __restrict__ lens,
                                                           int32_t S, int32_t* __restrict__ out_lcp, int32_t* __restrict__ out_unsorted) {
  __shared__ int first_bad;
  const int pair = blockIdx.x;
  if (pair >= S - 1) return;
  const int64_t* a = tokens + starts[pair]; const int64_t* b = tokens + starts[pair + 1];
  const int la = lens[pair], lb = lens[pair + 1];
  const int n = la < lb ? la : lb;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) first_bad = n;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int my_first = n;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pos = base + u * 256 + tid;
      const bool ne = pos < n && a[pos] != b[pos];
      const unsigned long long bal = __ballot(ne);
      if (bal != 0ull) {
        const int cand = base + u * 256 + (tid - lane) + __builtin_ctzll(bal);
        my_first = cand < my_first ? cand : my_first;
      }
    }
    if (lane == 0 && my_first < n) atomicMin(&first_bad, my_first);
    __syncthreads();
    const bool found = first_bad < n;   // uniform: every thread reads the same LDS word after the barrier ...
    __syncthreads();                    // ... and nobody may publish the NEXT step's candidate before everyone has read this one
    if (found) break;
  }
  if (tid == 0) {
    const int c = first_bad;
    out_lcp[pair] = c;
    if (c < n && a[c] > b[c]) atomicAdd(out_unsorted, 1);
  }
}

// Single workgroup of 1024 threads; chunks of 1024 sequences with a running base.
__global__ __launch_bounds__(1024) void leafize_kernel(const int32_t* __restrict__ lens, const int32_t* __restrict__ lcp, int32_t S,
                                                       int32_t* __restrict__ leaf_pos, int32_t* __restrict__ leaf_lcp,
                                                       int32_t* __restrict__ seq_leaf, int32_t* __restrict__ out_M) {
  __shared__ int wave_cnt[16];
  __shared__ int running;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) running = 0;
  __syncthreads();
  for (int base = 0; base < S; base += 1024) {
    const int i = base + tid;
    bool keep = false;
    if (i < S) {
      if (i == S - 1) keep = true;
      else {
        const int li = lens[i], lj = lens[i + 1];
        keep = lcp[i] < (li < lj ? li : lj);
      }
    }
    const unsigned long long bal = __ballot(keep);
    const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __builtin_popcountll(bal);
    __syncthreads();
    int wave_base = running;
    for (int w = 0; w < wave; ++w) wave_base += wave_cnt[w];
    if (i < S) {
      const int m = wave_base + before;
      seq_leaf[i] = m;
      if (keep) { leaf_pos[m] = i; if (i < S - 1) leaf_lcp[m] = lcp[i]; }
    }
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_cnt[w]; running += t; }
    __syncthreads();
  }
  if (tid == 0) *out_M = running;
}

__global__ __launch_bounds__(256) void preorder_meta_kernel(const int64_t* __restrict__ tokens, const int64_t* __restrict__ leaf_tok_off,
                                                            const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_depth0,
                                                            const int32_t* __restrict__ parent_of_seg,
                                                            const int32_t* __restrict__ brk_ptr, const int32_t* __restrict__ brk_depth,
                                                            const int32_t* __restrict__ brk_end, int32_t M, int32_t T,
                                                            int64_t* __restrict__ out_token, int32_t* __restrict__ out_depth,
                                                            int32_t* __restrict__ out_parent, int32_t* __restrict__ out_se) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
    int lo = 0, hi = M;                       // last segment with seg_off <= t
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_off[mid] <= t) lo = mid; else hi = mid; }
    const int i = lo, j = t - seg_off[i];
    const int d = seg_depth0[i] + j;
    out_token[t] = tokens[leaf_tok_off[i] + d];
    out_depth[t] = d;
    out_parent[t] = j > 0 ? t - 1 : parent_of_seg[i];
    int a = brk_ptr[i], b = brk_ptr[i + 1];   // last break with brk_depth <= d
    while (b - a > 1) { const int mid = (a + b) >> 1; if (brk_depth[mid] <= d) a = mid; else b = mid; }
    out_se[t] = brk_end[a];
  }
}

}  // namespace

extern "C" int dta_version(void) { return 200; }

extern "C" int dta_take_pending_error(char* msg, int32_t cap) {
  const hipError_t e = hipGetLastError();            // returns AND clears the thread's pending error
  if (msg && cap > 0) {
    const char* s = e == hipSuccess ? "" : hipGetErrorString(e);
    int i = 0;
    for (; s[i] && i < cap - 1; ++i) msg[i] = s[i];
    msg[i] = 0;
  }
  return (int)e;
}

extern "C" int dta_lcp_adjacent(const int64_t* tokens, const int64_t* starts, const int32_t* lens, int32_t S,
                                int32_t* out_lcp, int32_t* out_unsorted, void* stream) {
  if (S < 0 || !out_unsorted || (S > 0 && (!tokens || !starts || !lens)) || (S > 1 && !out_lcp)) return DTA_EINVAL;
  if (S <= 1) return DTA_OK;                       // an empty batch or a single sequence has no adjacent pair
  DTA_REFUSE_IF_PRIOR_ERROR();
  hipLaunchKernelGGL(lcp_adjacent_kernel, dim3(S - 1), dim3(256), 0, static_cast<hipStream_t>(stream), tokens, starts, lens, S, out_lcp, out_unsorted);
  return DTA_LAUNCH_STATUS();
}

extern "C" int dta_leafize(const int32_t* lens, const int32_t* lcp, int32_t S,
                           int32_t* out_leaf_pos, int32_t* out_leaf_lcp, int32_t* out_seq_leaf, int32_t* out_M, void* stream) {
  if (!lens || !out_leaf_pos || !out_leaf_lcp || !out_seq_leaf || !out_M || S < 1 || (S > 1 && !lcp)) return DTA_EINVAL;
  if (S > (1 << 20)) return DTA_EUNSUPPORTED;
  DTA_REFUSE_IF_PRIOR_ERROR();
  hipLaunchKernelGGL(leafize_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), lens, lcp, S, out_leaf_pos, out_leaf_lcp, out_seq_leaf, out_M);
  return DTA_LAUNCH_STATUS();
}

extern "C" int dta_preorder_meta(const int64_t* tokens, const int64_t* leaf_tok_off,
                                 const int32_t* seg_off, const int32_t* seg_depth0, const int32_t* parent_of_seg,
                                 const int32_t* brk_ptr, const int32_t* brk_depth, const int32_t* brk_end,
                                 int32_t M, int32_t T,
                                 int64_t* out_token, int32_t* out_depth, int32_t* out_parent, int32_t* out_subtree_end, void* stream) {
  if (!tokens || !leaf_tok_off || !seg_off || !seg_depth0 || !parent_of_seg || !brk_ptr || !brk_depth || !brk_end ||
      !out_token || !out_depth || !out_parent || !out_subtree_end || M < 1 || T < 1) return DTA_EINVAL;
  int blocks = (T + 255) / 256; if (blocks > 2048) blocks = 2048;
  DTA_REFUSE_IF_PRIOR_ERROR();
  hipLaunchKernelGGL(preorder_meta_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), tokens, leaf_tok_off, seg_off, seg_depth0,
                     parent_of_seg, brk_ptr, brk_depth, brk_end, M, T, out_token, out_depth, out_parent, out_subtree_end);
  return DTA_LAUNCH_STATUS();
}
