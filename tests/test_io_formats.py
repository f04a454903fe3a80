"""Interchange formats: round trips and the recorded reference table (grad/…bf16.txt layout)."""
import os

import torch

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
