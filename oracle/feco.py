"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- FeCo (reference defense/feature_level.py:168-217).

Two parts with different status:
  * ``kmeans_ids``: the DETERMINISM CONTRACT of this repository's k-means (csrc/k_feco.hip header, version 2: the
    assignment is a contraction on the matrix pipes), restated in numpy float32 + the C fmaf chain of conv_chain.c.  PARITY UNPINNED by construction: the reference delegates clustering to libKMCUDA /
    kmeans_pytorch (neither installed) with a random initialisation, so there are no reference ids to compare with.
  * ``compress_from_ids``: the reference's own step after the ids (:204-216: per-cluster torch.mean, empty cluster
    i falls back to frame i when `force`, is skipped otherwise), in torch so autograd gives the reference gradient.
"""
import numpy as np
import torch


def _dpad(D):
    return 32 if D <= 32 else 64


def centre(x):
    """Contract step 1: mu[d] = (p_0 + ... + p_15) / F with p_q = sum of frames q, q + 16, ... (fp32); returns (x - mu, mu)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    F, D = x.shape
    part = np.zeros((16, D), dtype=np.float32)
    for q in range(16):
        s = np.zeros(D, dtype=np.float32)
        for i in range(q, F, 16):
            s = s + x[i]
        part[q] = s
    t = part[0].copy()
    for q in range(1, 16):
        t = t + part[q]
    mu = (t / np.float32(F)).astype(np.float32)
    return (x - mu).astype(np.float32), mu


def half_norms(c, dpad):
    """h_j = -|c_j|^2 / 2: squares (rounded) summed over the dpad padded dimensions by the butterfly s[d] += s[d ^ step]."""
    k, D = c.shape
    s = np.zeros((k, dpad), dtype=np.float32)
    s[:, :D] = c * c
    idx = np.arange(dpad)
    step = 1
    while step < dpad:
        s = s + s[:, idx ^ step]
        step *= 2
    return (np.float32(-0.5) * s[:, 0]).astype(np.float32)


def kmeans_ids(x, k, max_iter=10, init_frames=None):
    """x (F,D) float32 -> int32 ids (F,): contract version 2 of csrc/k_feco.hip.  Frames are centred; a frame goes to the
    centroid with the largest score h_j + x'.c_j (one float32 fmaf chain in the contraction kernels' k order:
    oracle/conv_chain.c sg_feco_scores), first maximum wins; centroids are float32 sums of their centred frames in
    ascending frame order divided by the count, empty clusters keep their centroid.  Centroid j starts at frame
    floor(j F / k), or at init_frames[j] (the seeded form: oracle.philox.feco_random_init)."""
    from .conv_chain import feco_scores
    xc, _ = centre(x)
    F, D = xc.shape
    dp = _dpad(D)
    xp = np.zeros((F, dp), dtype=np.float32)
    xp[:, :D] = xc
    c = xc[[int(j * F // k) for j in range(k)] if init_frames is None else [int(f) for f in init_frames]].copy()
    ids = np.full(F, -1, dtype=np.int32)
    for _ in range(max_iter):
        cp = np.zeros((k, dp), dtype=np.float32)
        cp[:, :D] = c
        new = np.argmax(feco_scores(xp, cp, half_norms(c, dp)), axis=1).astype(np.int32)  # first maximum
        if np.array_equal(new, ids):
            break
        ids = new
        for j in range(k):
            members = np.nonzero(ids == j)[0]
            if members.size:
                s = np.zeros(D, dtype=np.float32)
                for i in members:
                    s = s + xc[i]
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
