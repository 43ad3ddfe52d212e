"""CPU checks of oracle/feco.py: the k-means determinism contract's invariants and the reference's
means-and-fallback step (defense/feature_level.py:204-216) incl. the empty-cluster branches."""
import numpy as np
import torch

from oracle import feco


def test_contract_kmeans_is_deterministic_and_locally_optimal():
    rs = np.random.RandomState(0)
    x = rs.randn(60, 8).astype(np.float32)
    ids = feco.kmeans_ids(x, 30, max_iter=40)
    assert np.array_equal(ids, feco.kmeans_ids(x.copy(), 30, max_iter=40))
    assert ids.min() >= 0 and ids.max() < 30
    # converged: every frame sits with its nearest centroid (centroids = cluster means)
    c = np.stack([x[ids == j].mean(0) if (ids == j).any() else np.full(8, 1e9, np.float32) for j in range(30)])
    d = ((x[:, None, :] - c[None]) ** 2).sum(-1)
    assert (d[np.arange(60), ids] <= d.min(1) + 1e-4).all()


def test_compress_from_ids_branches():
    x = torch.arange(24, dtype=torch.float32).view(6, 4)
    ids = np.array([0, 0, 2, 2, 2, 0])  # cluster 1 is empty
    forced = feco.compress_from_ids(x, ids, 3, force=True)
    assert forced.shape == (3, 4)
    assert torch.equal(forced[0], x[[0, 1, 5]].mean(0)) and torch.equal(forced[1], x[1]) and torch.equal(forced[2], x[[2, 3, 4]].mean(0))
    dropped = feco.compress_from_ids(x, ids, 3, force=False)
    assert dropped.shape == (2, 4) and torch.equal(dropped[1], forced[2])
    y = feco.feco(torch.randn(2, 40, 5), param=0.5)
    assert y.shape == (2, 20, 5)
