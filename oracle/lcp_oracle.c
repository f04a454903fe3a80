/* ORACLE — test infrastructure only.  Plain-C restatement of the token-level integer steps for
 * full-size inputs (the numpy/Python oracle is used for small cases):
 *   lcp_adjacent  — token_trie.py:6-10 + the order check token_trie.py:24-30
 *   leafize_keep  — token_trie.py:39 (which sorted sequences stay leaves)
 * Used by tests (ctypes) to check dta_lcp_adjacent / dta_leafize at BASELINE sizes. */
#include <stdint.h>

int lcp_adjacent(const int64_t* tokens, const int64_t* starts, const int32_t* lens, int32_t S,
                 int32_t* out_lcp, int32_t* out_unsorted) {
  int bad = 0;
  for (int32_t i = 0; i + 1 < S; ++i) {
    const int64_t* a = tokens + starts[i];
    const int64_t* b = tokens + starts[i + 1];
    int32_t n = lens[i] < lens[i + 1] ? lens[i] : lens[i + 1], c = 0;
    while (c < n && a[c] == b[c]) ++c;
    if (c < n && a[c] > b[c]) ++bad;
    out_lcp[i] = c;
  }
  *out_unsorted = bad;
  return 0;
}

int leafize_keep(const int32_t* lens, const int32_t* lcp, int32_t S, int32_t* leaf_pos, int32_t* leaf_lcp, int32_t* seq_leaf) {
  int32_t m = 0;
  for (int32_t i = 0; i < S; ++i) {
    int keep = (i == S - 1) || lcp[i] < (lens[i] < lens[i + 1] ? lens[i] : lens[i + 1]);
    seq_leaf[i] = m;
    if (keep) { leaf_pos[m] = i; if (i < S - 1) leaf_lcp[m] = lcp[i]; ++m; }
  }
  return m;
}
