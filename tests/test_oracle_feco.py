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


FECO_CASES = ("mfcc_b3", "logmel_b2", "single_drop", "single_full")


def _feco_ref():
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "feco_ref.npz"))
    return z, json.loads(str(z["meta"]))


def test_compress_from_ids_reproduces_the_reference_run():
    """tests/golden/feco_ref.npz = the reference's own FEATURE_COMPRESSION / kmeans code (feature_level.py:21-50,168-217)
    run on given ids (its third-party k-means replaced by a placeholder that returns them): outputs incl. the
    empty-cluster fallback (batch) and drop (single utterance) branches, and autograd's gradient.  The restatement must
    reproduce them exactly -- same torch ops in the same order."""
    z, meta = _feco_ref()
    assert "placeholder" in meta["accommodation"] and len(meta["kmeans_calls"]) == 7
    for tag in FECO_CASES:
        feat = torch.from_numpy(z[tag + "_feat"]).clone().requires_grad_(True)
        ids, k = z[tag + "_ids"], int(z[tag + "_k"])
        B = feat.shape[0]
        assert k == int(feat.shape[1] * float(z[tag + "_ratio"]))  # :186
        y = torch.cat([feco.compress_from_ids(feat[b], ids[b], k, force=B > 1).unsqueeze(0) for b in range(B)], dim=0)
        want = torch.from_numpy(z[tag + "_out"])
        assert y.shape == want.shape, (tag, y.shape, want.shape)
        assert torch.equal(y.detach(), want), tag
        (y * torch.from_numpy(z[tag + "_cot"])).sum().backward()
        assert torch.equal(feat.grad, torch.from_numpy(z[tag + "_dfeat"])), tag
    # the branches the fixture was built to reach
    assert z["single_drop_out"].shape[1] == int(z["single_drop_k"]) - 3
    ids = z["mfcc_b3_ids"]
    assert not (ids[1] == 3).any() and np.array_equal(z["mfcc_b3_out"][1, 3], z["mfcc_b3_feat"][1, 3])  # fallback = frame i


def _lloyd_float64(x, k, max_iter=10):
    """Textbook Lloyd iterations in float64 with the contract's start (centroid j = frame floor(j F / k)), stop rule and
    empty-cluster rule: the literal 'nearest centroid by squared distance'."""
    x = x.astype(np.float64)
    F = x.shape[0]
    c = x[[int(j * F // k) for j in range(k)]].copy()
    ids = np.full(F, -1)
    for _ in range(max_iter):
        d = ((x[:, None, :] - c[None]) ** 2).sum(-1)
        new = d.argmin(1)
        if np.array_equal(new, ids):
            break
        ids = new
        for j in range(k):
            if (ids == j).any():
                c[j] = x[ids == j].mean(0)
    return ids


def test_contract_version_2_is_the_literal_nearest_centroid_clustering():
    """Round 5 turned the assignment into a contraction (largest x'.c - |c|^2 / 2 on centred frames instead of the
    smallest sum of squared differences): on generic data -- no exact ties -- the float32 fmaf-chain scores must pick the
    centroids the float64 distances pick, also when the features sit far from the origin (log-mel in dB, +300 offset)."""
    rs = np.random.RandomState(5)
    for name, x, k in (("randn 120x16", rs.randn(120, 16), 40), ("log-mel-like 300x32", rs.randn(300, 32) * 10 - 40, 150),
                       ("offset +300", rs.randn(200, 32) * 4 + 300, 100), ("mfcc-like 300x30", rs.randn(300, 30) * [20] + 3, 150),
                       ("wide 90x64", rs.randn(90, 64), 36)):
        x = np.asarray(x, dtype=np.float32)
        got, want = feco.kmeans_ids(x, k), _lloyd_float64(x, k)
        assert np.array_equal(got, want), "%s: %d of %d ids differ" % (name, int((got != want).sum()), len(got))
