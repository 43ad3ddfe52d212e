"""Generate tests/golden/feco_ref.npz (build container only): the REFERENCE's own FeCo code after the clustering.

    python tests/golden/make_golden_feco.py      # needs /root/reference

What is pinned: ``defense/feature_level.py`` ``FEATURE_COMPRESSION`` (:21-50) -> ``kmeans`` (:168-217) -- k = int(n * ratio)
(:186), the per-cluster ``torch.mean(feat[ids])`` (:209), the empty-cluster fallback ``feat[i:i+1]`` when the batch has more
than one utterance (``force``, :33, :210-211), the DROP of empty clusters for a single utterance (:212-216), the batch
concatenation (:41-49) -- and torch autograd's gradient of all of it (the "tricky way to make FeCo differentiable", :204).

What is NOT pinned and cannot be: the cluster ids.  The reference takes them from ``kmeans_pytorch.kmeans`` (:199; random
initial centres, not installed here) or ``libKMCUDA.kmeans_cuda`` (:193).  Harness accommodation, disclosed in the
fixture's ``meta``: a placeholder module ``kmeans_pytorch`` is put in ``sys.modules`` whose ``kmeans(X, num_clusters,
distance, device)`` checks its arguments and returns the NEXT PRECOMPUTED ``(cluster_ids, centers)`` from a queue this
script fills -- it performs no clustering.  Everything downstream of that return value is reference arithmetic, executed
unmodified.  The ids in the queue are arbitrary but realistic: this repository's contract k-means (oracle/feco.py) on the
same features, with some clusters emptied on purpose so that both fallback branches run.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import feco as ofeco  # noqa: E402  (only to produce plausible ids)

QUEUE = []   # [(ids int64 (n,), k)] consumed in call order
CALLS = []   # what the reference asked for


def _install_kmeans_placeholder():
    def kmeans(X, num_clusters, distance='euclidean', device=None, **kw):
        assert not kw, kw
        ids, k = QUEUE.pop(0)
        assert X.shape[0] == ids.shape[0] and num_clusters == k, (X.shape, ids.shape, num_clusters, k)
        CALLS.append({"n": int(X.shape[0]), "dim": int(X.shape[1]), "k": int(num_clusters), "distance": distance})
        centers = torch.zeros(num_clusters, X.shape[1])  # :199 discards them (`cluster_ids, _ = ...`)
        return torch.from_numpy(ids.astype(np.int64)), centers
    mod = types.ModuleType("kmeans_pytorch")
    mod.kmeans = kmeans
    sys.modules["kmeans_pytorch"] = mod


def _ids_for(x, k, empty):
    ids = ofeco.kmeans_ids(x, k, max_iter=10).astype(np.int64)
    for j in empty:  # empty cluster j: its frames join the next non-emptied cluster
        tgt = (j + 1) % k
        while tgt in empty:
            tgt = (tgt + 1) % k
        ids[ids == j] = tgt
    return ids


def main():
    assert not torch.cuda.is_available()  # :192 would take the libKMCUDA branch
    _install_kmeans_placeholder()
    sys.path.insert(0, REF)
    from defense.feature_level import FEATURE_COMPRESSION, FeCo  # the reference module, unmodified

    rs = np.random.RandomState(20260301)
    out = {}
    cases = [
        # tag, B, F, D, ratio, clusters to empty per utterance
        ("mfcc_b3", 3, 60, 30, 0.5, [[], [3, 17], [0, 29]]),          # force=True: fallback row = frame i (:210-211)
        ("logmel_b2", 2, 75, 32, 0.2, [[], [14]]),                      # AudioNet-like feature width, k = 15
        ("single_drop", 1, 48, 30, 0.5, [[2, 9, 23]]),                  # force=False: empty clusters are skipped -> (1, k-3, D)
        ("single_full", 1, 41, 30, 0.3, [[]]),                          # n * ratio not an integer: k = int(12.3) = 12
    ]
    for tag, B, F, D, ratio, empties in cases:
        feat_np = (rs.randn(B, F, D) * 3.0).astype(np.float32)
        k = int(F * ratio)
        ids = np.stack([_ids_for(feat_np[b], k, empties[b]) for b in range(B)])
        for b in range(B):
            QUEUE.append((ids[b], k))
        feat = torch.from_numpy(feat_np).clone().requires_grad_(True)
        y = FEATURE_COMPRESSION(feat, 'kmeans', ratio, 'L2') if tag != "logmel_b2" else FeCo(feat, param=ratio)
        cot = torch.from_numpy(rs.randn(*y.shape).astype(np.float32))
        (y * cot).sum().backward()
        assert not QUEUE
        out[tag + "_feat"] = feat_np
        out[tag + "_ids"] = ids.astype(np.int32)
        out[tag + "_k"] = np.int32(k)
        out[tag + "_ratio"] = np.float64(ratio)
        out[tag + "_out"] = y.detach().numpy()
        out[tag + "_cot"] = cot.numpy()
        out[tag + "_dfeat"] = feat.grad.numpy()
        print(tag, "feat", feat_np.shape, "k", k, "out", tuple(y.shape), "empty", empties)
    assert all(c["distance"] == "euclidean" for c in CALLS)
    meta = {
        "generator": "tests/golden/make_golden_feco.py",
        "reference": "SpeakerGuard @ 2024-12-20 (/root/reference) defense/feature_level.py FEATURE_COMPRESSION/kmeans, unmodified",
        "accommodation": "sys.modules['kmeans_pytorch'] = placeholder whose kmeans() returns precomputed (ids, centers) "
                         "from a queue and does no clustering; torch.cuda.is_available() is False so :198-200 is the branch taken",
        "ids_source": "oracle/feco.py kmeans_ids (this repository's contract) with listed clusters emptied by hand; "
                      "the reference's own ids (random init, third-party) are not reproducible",
        "kmeans_calls": CALLS,
        "torch": torch.__version__, "numpy": np.__version__,
    }
    path = os.path.join(HERE, "feco_ref.npz")
    np.savez_compressed(path, meta=json.dumps(meta), **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
