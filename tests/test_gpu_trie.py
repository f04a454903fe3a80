"""GPU parity (bit-exact): dta_lcp_adjacent / dta_leafize / dta_preorder_meta through the product's
TokenTrie and packed engine metadata, against the reference-generated fixtures, the oracle, and —
at BASELINE sizes — the C oracle and size-independent properties."""
import ctypes
import os

import numpy as np
import pytest
import torch

import cases

import hostmirror
from dynamictreeattn_amd import synth
from dynamictreeattn_amd.token_trie import TokenTrie, _DeviceTokens, _device_trie_arrays
from oracle import trie_oracle as to

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _attach_table(t):
    return [[[a["_sequence_batch_id"], n] for a, n in al] for al in t.attach_lists]


def test_tokentrie_all_fixture_cases_bit_exact(trie_golden):
    for item in trie_golden["cases"]:
        seqs = synth.as_tensors(synth.make_case(item["case"])); ref = item["ref"]
        t = TokenTrie(seqs)
        assert t.lens == ref["lens"] and t.lcp_lens == ref["lcp_lens"] and _attach_table(t) == ref["attach"], item["case"]
        for name in ("forward_permute", "backward_permute"):
            t2 = TokenTrie(synth.as_tensors(synth.make_case(item["case"]))); getattr(t2, name)()
            assert t2.lens == ref[name]["lens"] and t2.lcp_lens == ref[name]["lcp_lens"] and _attach_table(t2) == ref[name]["attach"]
            # permute() derives the new adjacent LCPs on the host (range minima of the sorted-order kernel output); the kernel
            # recomputation from the tokens (what token_trie.py:94 does) must give the same integers
            assert t2.lcp_from_tokens() == t2.lcp_lens, (item["case"], name)
        t3 = TokenTrie(synth.as_tensors(synth.make_case(item["case"])))
        t3.random_permute(); assert t3.lcp_from_tokens() == t3.lcp_lens
        t3.backward_permute(); assert t3.lcp_from_tokens() == t3.lcp_lens          # a permutation of a permuted trie
        assert sorted(map(tuple, (x.tolist() for x in t3.inputs))) == sorted(map(tuple, (x.tolist() for x in t.inputs)))


def test_unsorted_flag_raises_like_reference():
    with pytest.raises(ValueError, match="Input_ids not sorted in lexicographic order."):
        TokenTrie(synth.as_tensors([[5, 1], [3, 2]]), sorted=True)
    TokenTrie(synth.as_tensors([[3, 2], [5, 1]]), sorted=True)


def test_edge_cases():
    t = TokenTrie(synth.as_tensors([[7]]))
    assert (t.lens, t.lcp_lens, t.n_tokens) == ([1], [], 1)
    e = TokenTrie([])                                           # the reference returns an empty trie (token_trie.py:52-82)
    assert (e.inputs, e.attach_lists, e.lens, e.lcp_lens, e.n_sequences, e.n_tokens) == ([], [], [], [], 0, 0)
    assert e.get_stats("forward")["n_tree_tokens"] == 0
    with pytest.raises(ValueError):                             # as the reference: CompressedTrie([], []) has len(lcp) != len(lens) - 1 (trie.py:77-78)
        e.backward_permute()
    seqs = [[1] * 5000, [1] * 4999 + [2], [1] * 1023 + [3], [1] * 1024 + [4], [1] * 1025 + [0]]
    t = TokenTrie(synth.as_tensors(seqs))
    o = to.TokenTrieOracle([np.array(s) for s in seqs])
    assert t.lens == o.lens and t.lcp_lens == o.lcp_lens        # mismatches exactly at 1024-step boundaries


def test_full_size_lcp_vs_c_oracle():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liblcp_oracle.so"))
    dev = torch.device("cuda:0")
    for case in ({"kind": "wide", "seed": 1}, {"kind": "tau2", "seed": 5, "G": 16, "turns": 12, "lo": 400, "hi": 1800}):
        seqs = synth.as_tensors(synth.make_case(case))
        keys = [s.numpy().astype(">u8").tobytes() for s in seqs]
        order = sorted(range(len(seqs)), key=keys.__getitem__)
        d = _DeviceTokens(seqs, dev)
        lcp, unsorted, leaf_pos = _device_trie_arrays(d, order, True)
        S = len(seqs)
        toks = np.concatenate([s.numpy() for s in seqs]); starts = d.starts[order].astype(np.int64); lens = d.lens[order].astype(np.int32)
        out = np.zeros(S, np.int32); bad = np.zeros(1, np.int32)
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        lib.lcp_adjacent(P(toks), P(starts), P(lens), S, P(out), P(bad))
        lp = np.zeros(S, np.int32); ll = np.zeros(S, np.int32); sl = np.zeros(S, np.int32)
        M = lib.leafize_keep(P(lens), P(out), S, P(lp), P(ll), P(sl))
        assert unsorted == 0 and bad[0] == 0
        assert leaf_pos == lp[:M].tolist() and lcp == ll[:M - 1].tolist()
        # property: sorted order => lcp[i] tokens really agree and the next one does not
        for i in range(0, S - 1, max(1, S // 16)):
            a, b = seqs[order[i]], seqs[order[i + 1]]
            c = int(out[i]); assert torch.equal(a[:c], b[:c]) and (c == min(len(a), len(b)) or a[c] < b[c])


def test_preorder_meta_kernel_vs_host_mirror():
    from dynamictreeattn_amd.tree_training_engine import _PackedTrie
    dev = torch.device("cuda:0")
    for case in cases.trie_cases()[:14] + cases.trie_cases()[20:30]:
        seqs = synth.as_tensors(synth.make_case(case))
        for perm in ("forward_permute", "backward_permute"):
            t = TokenTrie(seqs); getattr(t, perm)()
            pk = _PackedTrie(t, dev)
            seg, depth, parent, se = hostmirror.expand_plan_host(pk.plan)
            assert pk.depth.cpu().numpy().tolist() == depth.tolist()
            assert pk.parent.cpu().numpy().tolist() == parent.tolist()
            assert pk.subtree_end.cpu().numpy().tolist() == se.tolist()
            # packed tokens: every leaf's root path spells the leaf
            tok = pk.tokens.cpu()
            for i, leaf in enumerate(t.inputs):
                assert torch.equal(tok[pk.paths[i].cpu()], leaf)
            assert int(pk.meta.ktile_qend.max()) == pk.plan.T
            # the host-planned key-tile query ends (no device->host read) equal the per-tile maximum of the kernel's subtree_end
            from dynamictreeattn_amd import ops
            assert torch.equal(pk.meta.ktile_qend.cpu(), ops.ktile_qend_from(pk.subtree_end).cpu())


def test_engine_refuses_a_trie_on_another_device_and_unsupported_dtypes():
    from dynamictreeattn_amd.tree_training_engine import TreeTrainingEngine, _PackedTrie
    from dynamictreeattn_amd import model as m
    t = TokenTrie(synth.as_tensors([[1, 2, 3], [1, 2, 4]]))
    with pytest.raises(RuntimeError, match="holds its tokens on"):
        _PackedTrie(t, torch.device("cuda", 1))                # no second GPU needed: the check is on the device ids
    cfg = m.make_config(synth.QWEN3_0P6B)
    with pytest.raises(TypeError, match="bfloat16 / torch.float16"):
        TreeTrainingEngine(cfg, "cuda:0", torch.float64, 128)
    TreeTrainingEngine(cfg, "cuda:0", torch.float32, 128)        # run.py:122-132 allows fp32: so does this engine (tests/test_gpu_fp32.py)
