"""A tiny differentiable speaker-scoring model used to pin the ATTACK logic only.

It follows the reference model protocol (README.md:163: ``make_decision(x) -> (decisions,
scores)``, ``.threshold``) and is deterministic, so that running the reference attack classes
against it (tests/golden/make_golden.py) and running any restatement against it can be
compared trajectory by trajectory.  It is not part of the product.
"""
import numpy as np
import torch


class ToyModel(torch.nn.Module):
    def __init__(self, T=800, n_spk=4, seed=7, threshold=None):
        super().__init__()
        rs = np.random.RandomState(seed)
        self.pool = 40
        self.w1 = torch.nn.Parameter(torch.tensor(rs.randn(T // self.pool, 16) * 3.0, dtype=torch.float32))
        self.w2 = torch.nn.Parameter(torch.tensor(rs.randn(16, n_spk) * 2.0, dtype=torch.float32))
        self.threshold = threshold if threshold is not None else -np.inf
        self.n_spk = n_spk

    def score(self, x):
        B = x.shape[0]
        h = x.view(B, -1, self.pool).mean(2) * 8.0
        h = torch.tanh(h.matmul(self.w1))
        return h.matmul(self.w2)

    def forward(self, x):
        return self.score(x)

    def make_decision(self, x):
        scores = self.score(x)
        decisions = torch.argmax(scores, dim=1)
        max_scores = torch.max(scores, dim=1)[0]
        decisions = torch.where(max_scores > self.threshold, decisions, torch.full_like(decisions, -1))
        return decisions, scores


def toy_inputs(B=4, T=800, seed=11):
    rs = np.random.RandomState(seed)
    x = np.clip(0.3 * rs.randn(B, 1, T), -0.9, 0.9).astype(np.float32)
    return torch.from_numpy(x)
