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


# ---- InvertedIndex::QueryBucket (src/inverted_index.h:62-75) --------------------------------------------------------------------------
def query_bucket(clustered, offsets, permutation, q, k, bucket, metric):
    """top-k of one shard for one query: TopN key = the row index in the clustered layout (lexicographic (dist, row), topn.h:12-29),
    ids through `permutation` afterwards.  Returns (dist, ids, rows), ascending."""
    rows = np.arange(int(offsets[bucket]), int(offsets[bucket + 1]), dtype=np.uint32)
    dist = dist_many(np.asarray(clustered, np.float32)[rows], q, metric)
    wd, wrow = topk_lex(dist, rows, k)
    return wd, np.asarray(permutation)[wrow], wrow


# ---- KMeansTreeRouter::Query (src/kmeans_tree_router.cpp:93-130) ------------------------------------------------------------------------
# The priority queue is std::priority_queue<PQEntry, vector, greater<>> whose comparison looks at `dist` only, so the pop order among
# equal keys is whatever libstdc++'s heap does.  _Heap below is that algorithm as published in bits/stl_heap.h (__push_heap :135-148,
# __adjust_heap :219-245, pop_heap = swap-out of the root + __adjust_heap), written independently of the C++ oracle and of the device
# code; kmtr_query(..., heap="heapq") is the same walk on Python's heapq with an insertion counter — a DIFFERENT heap that must agree
# whenever no two keys in the queue are equal.
class _Heap:
    def __init__(self):
        self.a = []   # (key, payload)

    @staticmethod
    def _comp(x, y):   # std::greater<PQEntry>: x > y on dist
        return x[0] > y[0]

    def _push_heap(self, hole, top, value):
        a = self.a
        parent = (hole - 1) // 2
        while hole > top and self._comp(a[parent], value):
            a[hole] = a[parent]
            hole = parent
            parent = (hole - 1) // 2
        a[hole] = value

    def push(self, value):
        self.a.append(value)
        self._push_heap(len(self.a) - 1, 0, value)

    def pop(self):
        a = self.a
        top = a[0]
        if len(a) > 1:
            value = a[-1]
            a[-1] = a[0]
            n = len(a) - 1
            hole = second = 0
            while second < (n - 1) // 2:
                second = 2 * (second + 1)
                if self._comp(a[second], a[second - 1]):
                    second -= 1
                a[hole] = a[second]
                hole = second
            if (n & 1) == 0 and second == (n - 2) // 2:
                second = 2 * (second + 1)
                a[hole] = a[second - 1]
                hole = second - 1
            self._push_heap(hole, 0, value)
        a.pop()
        return top

    def __len__(self):
        return len(self.a)


def kmtr_query(flat, q, budget, metric, heap="libstdcxx", record=None):
    """The walk on a flat tree dict (node_shard, node_cent_first, node_child_first, node_child_count, shard_root, centroids, S).
    Returns (min_dist float32[S], had_equal_keys).  `record`: a list that receives every evaluated (dist, shard) pair (FrequencyQuery)."""
    import heapq
    S = int(flat["S"])
    cent = np.asarray(flat["centroids"], np.float32)
    cf, chf, chc = flat["node_cent_first"], flat["node_child_first"], flat["node_child_count"]
    min_dist = np.full(S, np.finfo(np.float32).max, np.float32)
    lowest = np.float32(-np.finfo(np.float32).max)
    budget = int(np.int32(budget))
    std = _Heap()
    hq, counter = [], 0
    for b in range(S):   # roots enter with key lowest() (centroids_in_roots is false), :103-110
        u = int(flat["shard_root"][b])
        if heap == "heapq":
            heapq.heappush(hq, (lowest, counter, u))
            counter += 1
        else:
            std.push((lowest, u))
    equal = False
    while (len(hq) if heap == "heapq" else len(std)) and budget > 0:
        if heap == "heapq":
            key, _, u = heapq.heappop(hq)
            equal = equal or (len(hq) > 0 and hq[0][0] == key and key != lowest)
        else:
            key, u = std.pop()
        n = int(cf[u + 1]) - int(cf[u])
        budget -= n
        if n == 0:
            continue
        dist = dist_many(cent[int(cf[u]):int(cf[u + 1])], q, metric)
        b = int(flat["node_shard"][u])
        for i in range(n):
            min_dist[b] = min(min_dist[b], dist[i])
            if record is not None:
                record.append((dist[i], b))
            if i < int(chc[u]):
                child = int(chf[u]) + i
                if heap == "heapq":
                    heapq.heappush(hq, (dist[i], counter, child))
                    counter += 1
                else:
                    std.push((dist[i], child))
    return min_dist, equal


def probe_order(min_dist):
    """the final order when all values differ (any correct sort gives the reference's permutation, kmeans_tree_router.cpp:125-127)"""
    return np.argsort(min_dist, kind="stable").astype(np.int32)
