"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- FeCo (reference defense/feature_level.py:168-217).

Two parts with different status:
  * ``kmeans_ids``: the DETERMINISM CONTRACT of this repository's k-means (csrc/k_feco.hip header), restated in
    numpy float32.  PARITY UNPINNED by construction: the reference delegates clustering to libKMCUDA /
    kmeans_pytorch (neither installed) with a random initialisation, so there are no reference ids to compare with.
  * ``compress_from_ids``: the reference's own step after the ids (:204-216: per-cluster torch.mean, empty cluster
    i falls back to frame i when `force`, is skipped otherwise), in torch so autograd gives the reference gradient.
"""
import numpy as np
import torch


def kmeans_ids(x, k, max_iter=10, init_frames=None):
    """x (F,D) float32 -> int32 ids (F,).  Sequential-in-d float32 distances, first minimum wins, centroids are
    float32 sums in ascending frame order divided by the count, empty clusters keep their centroid.  Centroid j starts
    at frame floor(j F / k), or at init_frames[j] (the seeded form: oracle.philox.feco_random_init)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    F, D = x.shape
    c = x[[int(j * F // k) for j in range(k)] if init_frames is None else [int(f) for f in init_frames]].copy()
    ids = np.full(F, -1, dtype=np.int32)
    for _ in range(max_iter):
        acc = np.zeros((F, k), dtype=np.float32)
        for d in range(D):
            df = x[:, d, None] - c[None, :, d]
            acc = acc + df * df
        new = np.argmin(acc, axis=1).astype(np.int32)  # first minimum
        if np.array_equal(new, ids):
            break
        ids = new
        for j in range(k):
            members = np.nonzero(ids == j)[0]
            if members.size:
                s = np.zeros(D, dtype=np.float32)
                for i in members:
                    s = s + x[i]
                c[j] = s / np.float32(members.size)
    return ids


def compress_from_ids(feat, cluster_ids, k, force):
    """feature_level.py:204-216 for one utterance: feat (F,D) torch tensor -> (k' <= k, D)."""
    rows = []
    for i in range(k):
        ids = np.argwhere(cluster_ids == i).flatten()
        if ids.size > 0:
            rows.append(torch.mean(feat[ids, :], dim=0).unsqueeze(0))
        elif force:
            rows.append(feat[i:i + 1, :])
    return torch.cat(rows, dim=0)


def feco(feat, param=0.5, max_iter=10):
    """FEATURE_COMPRESSION (:19-50) with the contract k-means: (B,F,D) -> (B,k,D)."""
    out = []
    for x in feat:
        k = int(x.shape[0] * param)
        ids = kmeans_ids(x.detach().numpy(), k, max_iter)
        out.append(compress_from_ids(x, ids, k, force=feat.shape[0] > 1).unsqueeze(0))
    return torch.cat(out, dim=0)
