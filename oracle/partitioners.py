"""CPU restatement of the reference's k-means shard builders, for the tests only (oracle/ is test infrastructure: nothing the
product ships imports it).  Control flow restated in Python over the C oracle's k-means / router trainer / distance
(oracle.cpp); each function cites the reference lines it follows.  Small inputs only: the loops are plain Python.

Parity note: the reference sorts the flattened ratings with parlay::sort_inplace (overlapping_partitioning.cpp:201), whose order
for equal distances is not specified and parlay is not part of the checkout; equal distances keep their (point, rank) order here.
"""
import math

import numpy as np

FLT_MAX = np.float32(np.finfo(np.float32).max)


def recursive_kmeans_partitioning(orc, points, max_cluster_size, depth=0, num_clusters=-1, metric=0):
    """src/partitioning.cpp:12-88"""
    n = len(points)
    if num_clusters < 0:
        num_clusters = int(math.ceil(float(n) / max_cluster_size))
    if num_clusters == 0:
        return np.zeros(n, np.int32)
    partition, _ = orc.kmeans(points, num_clusters, 555, metric)   # RandomSample(555) + KMeans (:19-26)
    partition = partition.copy()
    num_clusters = int(partition.max()) + 1
    sizes = np.bincount(partition, minlength=num_clusters)
    next_id = num_clusters
    for part in range(num_clusters):   # :49-85
        if sizes[part] <= max_cluster_size:
            continue
        members = np.flatnonzero(partition == part)
        sub = recursive_kmeans_partitioning(orc, points[members], max_cluster_size, depth + 1, -1, metric)
        moved = sub != 0
        partition[members[moved]] = next_id + sub[moved] - 1
        next_id += int(sub.max())
    return partition


def kmeans_partitioning(orc, points, num_clusters, epsilon, metric=0):
    """src/partitioning.cpp:141-144"""
    max_cluster_size = int(len(points) * (1 + epsilon) / num_clusters)
    return recursive_kmeans_partitioning(orc, points, max_cluster_size, 0, num_clusters, metric)


def rebalancing_kmeans_partitioning(orc, points, max_cluster_size, num_clusters=-1, metric=0):
    """src/partitioning.cpp:90-139"""
    n = len(points)
    if num_clusters < 0:
        num_clusters = int(math.ceil(float(n) / max_cluster_size))
    if num_clusters == 0:
        return np.zeros(n, np.int32)
    partition, centroids = orc.kmeans(points, num_clusters, 555, metric)
    partition = partition.copy()
    num_clusters = int(partition.max()) + 1
    sizes = np.bincount(partition, minlength=num_clusters)
    if not (sizes > max_cluster_size).any():
        return partition
    clusters = [list(np.flatnonzero(partition == c)) for c in range(num_clusters)]
    for c in range(num_clusters):   # :117-137, one point at a time
        while len(clusters[c]) > max_cluster_size:
            v = clusters[c][-1]
            dist = orc.distance_many(centroids, points[v], metric)   # distance(point, centroid) argument order is symmetric bitwise
            best, target = FLT_MAX, -1
            for j in range(len(clusters)):
                if len(clusters[j]) < max_cluster_size and dist[j] < best:
                    best, target = dist[j], j
            assert target != -1
            clusters[target].append(v)
            partition[v] = target
            clusters[c].pop()
    return partition


def make_overlapping_with_centroids(orc, points, clusters, max_cluster_size, num_extra_assignments, metric=0):
    """src/overlapping_partitioning.cpp:143-227; `clusters` is a list of id lists, returned extended"""
    clusters = [list(map(int, c)) for c in clusters if len(c) > 0]
    S = len(clusters)
    cover = [[] for _ in range(len(points))]
    for c, ids in enumerate(clusters):   # ConvertClustersToCover
        for u in ids:
            cover[u].append(c)
    tree = orc.kmtr_train(points, [np.asarray(c, np.uint32) for c in clusters], 64, 350, S * 64, metric)   # :151-158
    flat = tree.flat()
    reps, rep_cluster = flat["centroids"], tree.extract_partition()
    ratings = []
    for u in range(len(points)):   # :170-196
        dist = orc.distance_many(reps, points[u], metric)
        min_dist = np.full(S, FLT_MAX, np.float32)
        for j in range(len(reps)):
            if dist[j] < min_dist[rep_cluster[j]]:
                min_dist[rep_cluster[j]] = dist[j]
        for c in cover[u]:
            min_dist[c] = FLT_MAX
        targets = [c for c in range(S) if min_dist[c] != FLT_MAX and len(clusters[c]) < max_cluster_size]
        targets.sort(key=lambda c: min_dist[c])
        for t in targets[:5]:
            ratings.append((min_dist[t], t, u))
    order = sorted(range(len(ratings)), key=lambda i: ratings[i][0])   # stable
    left = int(num_extra_assignments)
    for i in order:   # :209-221; the reference's counter is unsigned and is tested only after a decrement-or-skip
        _, t, u = ratings[i]
        if len(clusters[t]) < max_cluster_size:
            left = (left - 1) % (1 << 64)
            clusters[t].append(u)
        if left == 0:
            break
    return clusters


def overlapping_kmeans_partitioning_spann(orc, points, partition, requested_num_clusters, epsilon, overlap, metric=0):
    """src/overlapping_partitioning.cpp:229-235"""
    n = len(points)
    num_extra = int(overlap * n)
    max_cluster_size = int((1.0 + epsilon) * n / requested_num_clusters)
    S = int(np.max(partition)) + 1
    clusters = [list(np.flatnonzero(np.asarray(partition) == c)) for c in range(S)]
    return make_overlapping_with_centroids(orc, points, clusters, max_cluster_size, num_extra, metric)
