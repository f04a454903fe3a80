"""Interchange formats: round trips and the recorded reference table (grad/…bf16.txt layout)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dynamictreeattn_amd import io


def test_batches_bins_stats_and_grad_tables(tmp_path):
    seqs = [torch.arange(5), torch.arange(3), torch.tensor([9, 8, 7, 6])]
    p = tmp_path / "call1.pt"
    torch.save(seqs, p)
    assert all(torch.equal(a, b) for a, b in zip(io.load_batch(str(p)), seqs))
    paths = io.save_bins(str(tmp_path / "bins"), "call1", seqs, [[0, 2], [1]])
    assert [os.path.basename(x) for x in paths] == ["call1_bin0.pt", "call1_bin1.pt"]
    assert [len(b) for _, b in io.load_folder(str(tmp_path / "bins"))] == [2, 1]
    rows = [{"name": "call1_bin0", "time": 2.0, "n_tokens": 9}, {"name": "call1_bin1", "time": 3.0, "n_tokens": 3},
            {"name": "call2", "time": 1.0, "n_tokens": 4}]
    io.write_stats(str(tmp_path / "s.jsonl"), rows)
    assert io.read_stats(str(tmp_path / "s.jsonl")) == rows
    assert io.dp_throughput(rows) == 16 / (3.0 + 1.0)                          # max over bins per call
    base = {"a": torch.ones(4), "b": torch.zeros(2), "c": None}
    exp = {"a": torch.ones(4) * 1.1, "b": torch.zeros(2), "d": torch.ones(1)}
    table = io.compare_grads(base, exp)
    assert [r[0] for r in table] == ["a", "b"] and abs(table[0][1] - 0.1) < 1e-6
    txt = io.format_compare_table(table)
    assert "Comparing 2 common parameters" in txt and "|Δg|/|g|" in txt


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_recorded_reference_table_round_trips_through_the_parser_and_formatter():
    """tests/golden/recorded_bf16_table.json = the reference's grad/Qwen3-0.6B-TB-vs-DB-bf16.txt as data (scripts/make_golden.py
    formats): re-formatting its rows gives the file's lines back and parsing those gives the rows — so a table written by this
    package is readable by the reference's tooling and the recorded one by ours."""
    rec = json.load(open(os.path.join(GOLD, "recorded_bf16_table.json")))
    rows = [tuple(r) for r in rec["rows"]]
    assert rec["n"] == len(rows) == 310
    assert abs(rec["max"] - 1.0636e-01) < 1e-9 and abs(rec["min"] - 8.4825e-03) < 1e-9 and rows[0][0] == "model.layers.1.self_attn.q_norm.weight"
    assert [r[1] for r in rows] == sorted((r[1] for r in rows), reverse=True)
    names = {r[0] for r in rows}
    assert len(names) == 310 and "model.embed_tokens.weight" in names and "lm_head.weight" not in names
    assert sum(n.startswith("model.layers.") for n in names) == 28 * 11
    text = io.format_compare_table(rows)
    assert text.split("\n")[:5] == rec["header_lines"]
    assert io.parse_compare_table(text) == rows


def test_compare_table_line_equal_to_the_reference_script_output():
    """exp/compare_grads.py run on two seeded dumps (fixture: its output text) vs io.compare_grads_report on the same dumps."""
    fx = json.load(open(os.path.join(GOLD, "formats_cases.json")))["compare"]
    g = torch.Generator().manual_seed(fx["seed"])
    names = fx["names"]
    base = {n: torch.randn(6, 5, generator=g) for n in names}
    exp = {n: base[n] + 0.01 * (i + 1) * torch.randn(6, 5, generator=g) for i, n in enumerate(names)}
    base["only.in.base"] = torch.ones(3); exp["only.in.exp"] = torch.ones(2); base["none.entry"] = None; exp["none.entry"] = torch.ones(1)
    assert io.compare_grads_report(base, exp).split("\n") == fx["table"].split("\n")


def test_stats_rows_carry_the_reference_keys_and_feed_the_time_model(monkeypatch):
    import hostmirror
    from dynamictreeattn_amd.token_trie import TokenTrie
    from dynamictreeattn_amd.tree_time_model import TreeTimeModel
    hostmirror.install(monkeypatch)
    fx = json.load(open(os.path.join(GOLD, "formats_cases.json")))
    seqs = [torch.tensor(s) for s in ([1, 2, 3, 4], [1, 2, 5], [1, 2])]
    t = TokenTrie(seqs, device="cpu")
    fwd = io.stats_row(t, "tree_forward", 0.0, "x")
    t.backward_permute()
    bwd = io.stats_row(t, "tree_backward", 0.0, "x", loss=0.0, block_size=1)
    assert list(fwd) == fx["stats_keys"]["tree_forward"] and list(bwd) == fx["stats_keys"]["tree_backward"]
    assert fwd == fx["stats_example"]["tree_forward"] and bwd == fx["stats_example"]["tree_backward"]      # values too
    # online fit (tree_time_model.py:14-39): rows in this format train the model once 16 of them are in
    tm = TreeTimeModel()
    rows = [dict(bwd, n_tree_tokens=100 + 10 * i, sum_depth=5000 + 700 * i * i, time=0.01 * (100 + 10 * i) + 1e-6 * (5000 + 700 * i * i)) for i in range(16)]
    tm.add_data(rows)
    assert tm.coeffs is not None and abs(tm.pred(rows[3]) - rows[3]["time"]) < 1e-3 * rows[3]["time"] + 1e-6


def test_partition_folder_writes_the_reference_bin_files(tmp_path, monkeypatch, balancer_golden):
    """data_parallel.py:177-199: `{name}_bin{k}.pt` per batch of a folder; bins equal the reference's recorded ones."""
    import hostmirror
    from dynamictreeattn_amd import data_parallel as lb, synth
    hostmirror.install(monkeypatch)
    from dynamictreeattn_amd import token_trie
    orig = token_trie.TokenTrie.__init__
    monkeypatch.setattr(token_trie.TokenTrie, "__init__", lambda self, *a, **kw: orig(self, *a, **{**kw, "device": torch.device("cpu")}))
    item = balancer_golden["cases"][4]       # wide trie: the reference does not raise there at block 64
    seqs = synth.as_tensors(synth.make_case(item["case"]))
    (tmp_path / "data").mkdir()
    torch.save(seqs, tmp_path / "data" / "callA.pt")
    res = lb.partition_folder(str(tmp_path / "data"), str(tmp_path / "out"), "LB_by_DFS_and_TM", 4, "backward", 64)
    exp = item["bins"]["K4_backward_64_raw"]["LB_by_DFS_and_TM"]
    assert res["bins"]["callA"] == exp
    files = sorted(os.listdir(tmp_path / "out"))
    assert files == [f"callA_bin{k}.pt" for k in range(4)]
    for k in range(4):
        got = io.load_batch(str(tmp_path / "out" / f"callA_bin{k}.pt"))
        assert len(got) == len(exp[k]) and all(torch.equal(a, seqs[i]) for a, i in zip(got, exp[k]))
    # forward mode drops the block size (data_parallel.py:155-156)
    res_f = lb.partition_folder(str(tmp_path / "data"), None, "LB_by_TM", 2, "forward", 2048)
    assert res_f["bins"]["callA"] == item["bins"]["K2_forward_None_raw"]["LB_by_TM"]
