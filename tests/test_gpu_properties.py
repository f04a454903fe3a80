"""GPU property tests (hypothesis): random rollouts through the HIP trie kernels and the HIP tree attention, against the oracle.
Few examples per test — every example is a handful of launches — but drawn fresh over sequence counts, lengths, duplicates,
prefix-of-each-other pairs and head geometries that the fixed cases of test_gpu_trie.py / test_gpu_attention.py do not enumerate."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

import test_gpu_attention as ga
from dynamictreeattn_amd.token_trie import TokenTrie
from oracle import trie_oracle as to

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])

rollouts = st.lists(st.lists(st.integers(0, 3), min_size=1, max_size=70), min_size=1, max_size=14)


@settings(max_examples=40, **COMMON)
@given(rollouts)
def test_hip_tokentrie_equals_the_oracle(seqs):
    t = TokenTrie([torch.tensor(s, dtype=torch.int64) for s in seqs], [{"i": i} for i in range(len(seqs))], device=DEV)
    o = to.TokenTrieOracle([np.asarray(s, dtype=np.int64) for s in seqs], [{"i": i} for i in range(len(seqs))])
    for step in ("sorted", "forward", "backward"):
        if step != "sorted":
            getattr(t, step + "_permute")(); getattr(o, step + "_permute")()
        assert [x.tolist() for x in t.inputs] == [x.tolist() for x in o.inputs], step
        assert list(t.lens) == list(o.lens) and list(t.lcp_lens) == list(o.lcp_lens), step
        assert [[(a["_sequence_batch_id"], n) for a, n in al] for al in t.attach_lists] == \
               [[(a["_sequence_batch_id"], n) for a, n in al] for al in o.attach_lists], step


@settings(max_examples=12, **COMMON)
@given(rollouts, st.sampled_from([(2, 1), (4, 2), (6, 2), (8, 8)]), st.sampled_from(["forward", "backward"]), st.integers(0, 3))
def test_hip_tree_attention_equals_the_oracle_on_random_tries(seqs, heads, order, seed):
    plan, se = ga._trie_case(seqs, order)
    ga._run(plan, se, heads[0], heads[1], torch.bfloat16, seed=seed)
