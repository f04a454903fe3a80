"""Linear cost model of one tree pass, fitted online by non-negative least squares.
Host-side, microseconds; kept because the balancers call ``.pred`` (tree_time_model.py:5-48).
T = c0·n_leaf_sequences + c1·n_tree_tokens + c2·n_f1_tokens + c3·sum_prefix_len + c4·sum_depth."""
from __future__ import annotations

import numpy as np

_FEATURES = ("n_leaf_sequences", "n_tree_tokens", "n_f1_tokens", "sum_prefix_len", "sum_depth")


class TreeTimeModel:
    MIN_N_DATA_POINTS = 16
    MAX_N_DATA_POINTS = 1024

    def __init__(self):
        self.coeffs = None
        self.data = []

    @staticmethod
    def _row(stats):
        return [stats.get(k, 0) if k == "n_f1_tokens" else stats[k] for k in _FEATURES]

    def fit(self):
        from scipy.optimize import nnls
        X = np.array([self._row(s) for s in self.data], dtype=np.float64)
        y = np.array([s["time"] for s in self.data], dtype=np.float64)
        self.coeffs, _ = nnls(X, y)
        return float(np.mean((X @ self.coeffs - y) ** 2))

    def add_data(self, data):
        self.data = (self.data + list(data))[-self.MAX_N_DATA_POINTS:]
        if len(self.data) >= self.MIN_N_DATA_POINTS:
            self.fit()

    def pred(self, stats):
        if self.coeffs is None:                       # untrained: tree tokens (tree_time_model.py:41-43)
            return stats["n_tree_tokens"]
        r = self._row(stats)
        return (self.coeffs[0] * r[0] + self.coeffs[1] * r[1] + self.coeffs[2] * r[2] + self.coeffs[3] * r[3] + self.coeffs[4] * r[4])
