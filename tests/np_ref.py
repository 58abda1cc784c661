"""Independent numpy float32 emulation of the reference's scalar distance sequences (src/dist.cpp:17-37, :42-52).
Used to cross-check the C++ oracle: every operation is a separate np.float32 op, so each rounds to binary32."""
import numpy as np


def l2_many(points, q):
    """points [n,d] f32, q [d] f32 -> [n] f32 with the 4-unrolled order: result += ((d0^2+d1^2)+d2^2)+d3^2, then the tail."""
    points = np.asarray(points, np.float32)
    q = np.asarray(q, np.float32)
    n, d = points.shape
    res = np.zeros(n, np.float32)
    i = 0
    while i + 4 <= d:
        diff = points[:, i:i + 4] - q[None, i:i + 4]          # a[j] - b[j], a = point
        sq = diff * diff
        s = sq[:, 0] + sq[:, 1]
        s = s + sq[:, 2]
        s = s + sq[:, 3]
        res = res + s
        i += 4
    while i < d:
        diff = points[:, i] - q[i]
        res = res + diff * diff
        i += 1
    return res


def mips_many(points, q):
    points = np.asarray(points, np.float32)
    q = np.asarray(q, np.float32)
    n, d = points.shape
    res = np.zeros(n, np.float32)
    for i in range(d):
        res = res + q[i] * points[:, i]
    return np.float32(1.0) - res


def dist_many(points, q, metric):
    return mips_many(points, q) if metric == 1 else l2_many(points, q)


def topk_lex(dists, ids, k):
    """k lexicographically smallest (dist, id), ascending"""
    order = np.lexsort((ids, dists))[:k]
    return dists[order], ids[order]
