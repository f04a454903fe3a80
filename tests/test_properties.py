"""Property tests (hypothesis) of the trie and packing host logic on random rollouts — SURVEY §4 asks for them because the
reference has no test suite of its own.  The three device steps run as tests/hostmirror.py stand-ins (CPU); the invariants are
checked against the oracle's restatement of the reference (tests/ may import oracle/) and against brute force on the token level."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

import hostmirror
from dynamictreeattn_amd import packing
from dynamictreeattn_amd.token_trie import TokenTrie
from dynamictreeattn_amd.trie import CompressedTrie, _get_stats
from oracle import trie_oracle as to

CPU = torch.device("cpu")
COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])

# few symbols and short sequences: many shared prefixes, duplicates and sequences that are prefixes of others
rollouts = st.lists(st.lists(st.integers(0, 2), min_size=1, max_size=24), min_size=1, max_size=10)


@pytest.fixture(autouse=True)
def _hooks(monkeypatch):
    hostmirror.install(monkeypatch)


def _tensors(seqs):
    return [torch.tensor(s, dtype=torch.int64) for s in seqs]


def _token_tree(leaves):
    """Brute force: the stack walk token by token -> (parent, depth, ancestors-or-self sets) in DFS pre-order of `leaves`."""
    par, dep, stack, prev = [], [], [], None
    for leaf in leaves:
        c = 0
        if prev is not None:
            while c < min(len(prev), len(leaf)) and prev[c] == leaf[c]:
                c += 1
        stack = stack[:c]
        for d in range(c, len(leaf)):
            par.append(stack[-1] if stack else -1); dep.append(d); stack.append(len(par) - 1)
        prev = leaf
    anc = []
    for x in range(len(par)):
        s, q = set(), x
        while q != -1:
            s.add(q); q = par[q]
        anc.append(s)
    return par, dep, anc


@settings(max_examples=120, **COMMON)
@given(rollouts)
def test_tokentrie_equals_the_oracle_and_keeps_every_sequence(seqs):
    attach = [{"tag": i} for i in range(len(seqs))]
    t = TokenTrie(_tensors(seqs), [dict(a) for a in attach], device=CPU)
    o = to.TokenTrieOracle([np.asarray(s, dtype=np.int64) for s in seqs], [dict(a) for a in attach])
    assert [x.tolist() for x in t.inputs] == [x.tolist() for x in o.inputs]
    assert list(t.lens) == list(o.lens) and list(t.lcp_lens) == list(o.lcp_lens)
    # every input sequence is the prefix (of its recorded length) of exactly the leaf it is attached to
    seen = {}
    for leaf, al in zip(t.inputs, t.attach_lists):
        for a, n in al:
            seen[a["_sequence_batch_id"]] = leaf.tolist()[:n]
    assert seen == {i: list(s) for i, s in enumerate(seqs)}
    # leaves are strictly increasing and none is a prefix of the next (that is what "leafization" means)
    L = [x.tolist() for x in t.inputs]
    for a, b in zip(L[:-1], L[1:]):
        assert a < b and b[:len(a)] != a
    for mode in ("forward", "backward"):
        assert t.get_stats(mode, 8) == o.get_stats(mode, 8)
    t.backward_permute(); o.backward_permute()
    assert [x.tolist() for x in t.inputs] == [x.tolist() for x in o.inputs] and list(t.lcp_lens) == list(o.lcp_lens)


@settings(max_examples=120, **COMMON)
@given(rollouts, st.sampled_from(["forward", "backward", "random"]))
def test_dfs_orders_are_permutations_with_consistent_prefix_lengths(seqs, which):
    t = TokenTrie(_tensors(seqs), device=CPU)
    ct = CompressedTrie(t.lens, t.lcp_lens)
    n = len(t.lens)
    if which == "random":
        order, lens, lcps = ct.get_order_random(seed=3), None, None
    else:
        order, lens, lcps = ct.get_order_forward() if which == "forward" else ct.get_order_backward()
    assert sorted(order) == list(range(n))
    L = [t.inputs[i].tolist() for i in order]
    # a DFS order keeps every subtree contiguous: the leaves below any prefix form one block
    for p in {tuple(x[:d + 1]) for x in L for d in range(len(x))}:
        pos = [i for i, x in enumerate(L) if tuple(x[:len(p)]) == p]
        assert pos == list(range(pos[0], pos[0] + len(pos)))
    if lens is None:
        return
    assert list(lens) == [len(x) for x in L]
    for a, b, c in zip(L[:-1], L[1:], lcps):                       # the lcp the trie reports is the real common prefix
        k = 0
        while k < min(len(a), len(b)) and a[k] == b[k]:
            k += 1
        assert k == c
    # any DFS order visits the same tree: the number of distinct tree tokens does not depend on the order
    st_ = _get_stats(lens, lcps, "forward", None)
    assert st_["n_tree_tokens"] == sum(lens) - sum(lcps) == len({tuple(x[:d + 1]) for x in L for d in range(len(x))})


@settings(max_examples=80, **COMMON)
@given(rollouts, st.sampled_from([4, 16]))
def test_packed_plan_and_tile_runs_cover_exactly_the_ancestors(seqs, tile):
    t = TokenTrie(_tensors(seqs), device=CPU); t.backward_permute()
    leaves = [x.tolist() for x in t.inputs]
    par, dep, anc = _token_tree(leaves)
    T = len(par)
    plan = packing.plan_segments(t.lens, t.lcp_lens)
    assert plan.T == T
    _, depth, parent, se = hostmirror.expand_plan_host(plan)
    assert depth.tolist() == dep and parent.tolist() == par
    # the ancestor test of the kernels: s is an ancestor-or-self of x  <=>  s <= x < subtree_end[s]
    for s in range(T):
        assert {x for x in range(T) if s <= x < se[s]} == {x for x in range(T) if s in anc[x]}
    # every leaf's path spells the leaf, root first
    for i, runs in enumerate(plan.path_runs):
        own = list(range(plan.seg_off[i], plan.seg_off[i + 1]))
        path = [x for b, e in runs for x in range(b, e)] + own
        assert [dep[x] for x in path] == list(range(len(leaves[i]))) and all(par[b] == a for a, b in zip(path[:-1], path[1:]))
    rp, runs = packing.plan_qtile_runs(plan, tile)
    for qt in range(len(rp) - 1):
        rows = range(qt * tile, min(T, qt * tile + tile))
        need = set().union(*(anc[x] for x in rows))
        cov = set()
        for b, e, f, _ in runs[rp[qt]:rp[qt + 1]]:
            assert 0 <= b < e <= T and not (cov & set(range(b, e)))               # runs are disjoint
            cov |= set(range(b, e))
            if f == 0:                                                            # unmasked run: every key is an ancestor of every row
                assert all(set(range(b, e)) <= anc[x] for x in rows)
        assert need <= cov
    # key-tile query ends planned on the host == the maximum subtree_end over the tile's keys
    kq = packing.ktile_qend_host(plan, tile)
    assert kq.tolist() == [int(se[k:k + tile].max()) for k in range(0, T, tile)]


@settings(max_examples=60, **COMMON)
@given(rollouts, st.sampled_from([16, 64]))
def test_padding_never_touches_real_tokens(seqs, mult):
    t = TokenTrie(_tensors(seqs), device=CPU); t.backward_permute()
    plan = packing.plan_segments(t.lens, t.lcp_lens)
    pp = packing.pad_plan(plan, mult)
    assert pp.T % mult == 0 and 0 <= pp.T - plan.T < mult and pp.n_real == plan.M
    _, d0, p0, s0 = hostmirror.expand_plan_host(plan)
    _, d1, p1, s1 = hostmirror.expand_plan_host(pp)
    T = plan.T
    assert d1[:T].tolist() == d0.tolist() and p1[:T].tolist() == p0.tolist() and s1[:T].tolist() == s0.tolist()
    assert all(p == -1 or p >= T for p in p1[T:].tolist())                          # filler hangs under no real token
