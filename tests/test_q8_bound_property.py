"""CPU property test of the 8-bit bound of the filter's sample / rest passes (gp-ann_amd/csrc/scan_cells.h: Q8Threshold, FilterKernelCells MODE 2
full8 / quick8; cells.hip: CellQ8ScaleKernel, CellQ8PlaneKernel, TileBoundKernel), restated in numpy:

    a row whose CANONICAL binary32 distance to the query is within tau passes the integer test — with the plain accumulators and with the
    accumulators biased by 1.5 x 2^23 (kernels for up to 128 dimensions) — and the test of a tile's first half (H.P' against the tile's
    largest norm word and 127 max ||P'||_1) lets through every tile that holds such a row.

The GPU parity tests check the kernels' RESULTS bit for bit; this one checks the bound's arithmetic itself, on data chosen to stress it: clouds far
from the origin, a cell with outliers (a coarse quantisation step), tiny and huge scales, thresholds that sit exactly on a row's distance."""
import numpy as np
import pytest

import np_ref

f32 = np.float32
B0 = 12582912.0            # 1.5 * 2^23: bit pattern 0x4B400000
K255 = 255.0 * B0          # a float: 765 * 2^22


def fma32(a, b, c):
    """fl32(a * b + c) for float32 arrays: the product of two binary32 numbers is exact in the 64-bit significand of np.longdouble"""
    return (np.asarray(a, np.longdouble) * np.asarray(b, np.longdouble) + np.asarray(c, np.longdouble)).astype(f32)


def up32(x):
    """smallest float32 >= x (x a python float)"""
    y = f32(x)
    return y if float(y) >= x else np.nextafter(y, f32(np.inf))


def q8_threshold(metric, tau, nq, q1, tscale, R, d, D_table, nc, s_c, radius, l1_table=-1.0):
    """scan_cells.h Q8Threshold, line by line (double arithmetic)"""
    if not tau < np.inf:
        return f32(-np.inf)
    qn = np.sqrt(nq) * (1.0 + 1e-12)
    if metric == 0:
        g1 = (d + 32.0) * 2.0 ** -23
        e2 = (2.0 ** -20 + 2.0 ** -21) * (R + qn) * (R + qn) + 1e-30
        Ub = -(tau * (1.0 + 2.0 * g1) - nq * (1.0 - 2.0 ** -40) + e2) * 0.5
    else:
        e1 = (d + 4.0) * 2.0 ** -23 * (qn * R + 1.0) + abs(tau) * 2.0 ** -22
        Ub = 1.0 - tau - e1 - 1e-30
    qc = 0.5 * (nq + nc - D_table)
    cn = np.sqrt(nc)
    e_tab = 0.5 * ((d + 8.0) * 2.0 ** -23 * abs(D_table) + 2.0 ** -21 * (qn + cn) * np.sqrt(abs(D_table)) + 2.0 ** -40 * (nq + nc))
    r_eff = radius + 2.0 ** -23 * (R + cn)          # |p~ - c^| <= |p - c| + the rounding of the centred coordinates
    # the own norm words of the 8-bit plane (squared L2): only (q~ - c^).e is unknown, |.| <= (s/2) |q~ - c^|_1
    if l1_table >= 0.0:
        q1_eff = l1_table * (1.0 + (d + 8.0) * 2.0 ** -24) + 2.0 ** -23 * np.sqrt(d) * (qn + cn)
    elif l1_table == -2.0:   # own norm words, no |q - c|_1 table: Cauchy-Schwarz on the squared distance
        q1_eff = np.sqrt(d * abs(D_table)) * (1.0 + (d + 8.0) * 2.0 ** -24) + 2.0 ** -23 * np.sqrt(d) * (qn + cn)
    else:
        q1_eff = q1
    E = 0.5 * s_c * (1.0 + 1e-6) * q1_eff + tscale / 508.0 * (1.0 + 1e-6) * (np.sqrt(d) * r_eff + 0.5 * d * s_c)
    e_fp = 2.0 ** -21 * (qn * (r_eff + np.sqrt(d) * s_c) + 0.5 * R * R + cn * np.sqrt(d) * s_c + 1.0)
    Bd0 = Ub - qc - E - e_tab - e_fp
    Bd = Bd0 - 1024.0 * s_c * tscale / 254.0 * (1.0 + 1e-6) - 2.0 ** -22 * abs(Bd0)
    Bf = f32(Bd)
    if float(Bf) > Bd:
        Bf = np.nextafter(Bf, f32(-np.inf))
    return Bf


def make_cell(rng, n, d, centre_offset, sigma, outliers):
    c_true = rng.standard_normal(d) * 1.0 + centre_offset
    pts = (c_true + sigma * rng.standard_normal((n, d))).astype(f32)
    if sigma == 0.0:
        return pts
    for _ in range(outliers):
        pts[rng.integers(0, n), rng.integers(0, d)] += f32(40.0 * sigma)
    return pts


@pytest.mark.parametrize("metric,cell_frame", [(0, True), (0, "cs"), (0, False), (1, False)])
@pytest.mark.parametrize("case", ["plain", "offset", "outliers", "tiny", "huge", "degenerate", "degenerate_ulp"])
def test_rows_within_tau_pass_the_8_bit_test(metric, cell_frame, case):
    """cell_frame (squared L2, the default store layout since round 4): the plane's own norm words -|p~|^2/2 + c^.e (cells.hip CellQ8NormKernel) with
    the error term (s/2) |q~ - c^|_1 from the fp32 table of |q - c|_1 (CellDistKernel<true>); else the plain norm words with (s/2) |q~|_1"""
    cases = ["plain", "offset", "outliers", "tiny", "huge", "degenerate", "degenerate_ulp"]
    rng = np.random.default_rng(1000 * metric + cases.index(case))
    d, n, nq = 48, 640, 24
    scale = {"tiny": 1e-3, "huge": 3e3}.get(case, 1.0)
    off = 100.0 if case == "offset" else 0.0
    degenerate = case.startswith("degenerate")
    if degenerate:
        # a cell of equal rows (radius 0 in the original coordinates), or of rows that differ by an ulp or two of their coordinates, FAR from
        # the store's mean: the quantisation step is tiny or the 1e-30 clamp, and |p~ - c^| is made of the roundings of the centred
        # coordinates alone — the term Q8Threshold carries explicitly (r_eff)
        base = (rng.standard_normal(d) + 100.0).astype(f32)
        pts = np.tile(base, (n, 1))
        if case == "degenerate_ulp":
            pts = (pts.astype(np.float64) + 1e-5 * rng.standard_normal((n, d))).astype(f32)   # (spacing of floats near 100: 7.6e-6)
        others = make_cell(rng, 2000, d, 0.0, 3.0, 0)                           # the rest of the store sits around the origin
        queries = (pts[rng.integers(0, n, nq)].astype(np.float64) + rng.choice([0.0, 1e-5, 0.3], nq)[:, None] * rng.standard_normal((nq, d))).astype(f32)
    else:
        pts = make_cell(rng, n, d, off, 0.35 * scale, 12 if case == "outliers" else 0)
        others = make_cell(rng, 2000, d, off, 3.0 * scale, 0)                       # the rest of the store: sets the frame and R
        queries = (pts[rng.integers(0, n, nq)] + (0.3 * scale * rng.standard_normal((nq, d))).astype(f32)).astype(f32)
    store = np.vstack([pts, others])
    # the frame (cells.hip: the store's mean row for squared L2, none for the inner product)
    mu = store.astype(np.float64).mean(axis=0).astype(f32) if metric == 0 else np.zeros(d, f32)
    pt = (pts - mu).astype(f32)                                                     # p~ = fl(p - mu)
    c = pts.astype(np.float64).mean(axis=0).astype(f32)                             # the cell's centre (any point will do)
    ct = (c - mu).astype(f32)
    res = (pt - ct).astype(f32)                                                     # the fp32 residuals the step is taken from
    s_c = up32(max(float(np.abs(res).max()) * (1.0 + 1e-6) / 127.0, 1e-30))
    Pq = np.clip(np.rint((pt.astype(np.float64) - ct.astype(np.float64)) / float(s_c)), -127, 127).astype(np.int64)
    nhn = (-0.5 * (pt.astype(np.float64) ** 2).sum(axis=1)).astype(f32) if metric == 0 else np.zeros(n, f32)
    if cell_frame:   # CellQ8NormKernel: -|p~|^2/2 + c^.e per row, e = (p~ - c^) - s P' (double), one rounding to a float
        e_q = (pt.astype(np.float64) - ct.astype(np.float64)) - float(s_c) * Pq
        nhn = (-0.5 * (pt.astype(np.float64) ** 2).sum(axis=1) + e_q @ ct.astype(np.float64)).astype(f32)
    # CellRadiusKernel: max |p - c| in the ORIGINAL coordinates (double), x (1 + 1e-6), rounded up to a float
    radius = float(up32(float(np.sqrt(((pts.astype(np.float64) - c.astype(np.float64)) ** 2).sum(axis=1)).max()) * (1.0 + 1e-6)))
    R = float(np.sqrt((((store - mu).astype(f32).astype(np.float64)) ** 2).sum(axis=1)).max()) * (1.0 + 1e-12)
    nc = float((ct.astype(np.float64) ** 2).sum())
    # tiles of 32 rows in the order of the norm words (RowsByNorm) — and in slab order: the tile test must hold either way
    for order in (np.argsort(-nhn, kind="stable"), np.arange(n)):
        for qi in range(nq):
            q = queries[qi]
            qt = (q - mu).astype(f32)
            nqd = float((qt.astype(np.float64) ** 2).sum())
            q1 = float(np.abs(qt.astype(np.float64)).sum())
            t_q = up32(max(float(np.abs(qt).max()) * (1.0 + 1e-6) / 127.0, 1e-30))
            u = qt.astype(np.float64) / float(t_q)
            H = np.clip(np.rint(u), -127, 127)
            L = np.clip(np.rint((u - H) * 254.0), -127, 127).astype(np.int64)
            H = H.astype(np.int64)
            E = np_ref.dist_many(pts, q, metric)                                    # canonical binary32 distances
            D_table, L1_table = f32(0.0), f32(0.0)
            for x in range(d):                                                      # CellDistKernel: plain fp32 sums of the squared / absolute fp32 differences
                df = f32(q[x] - c[x])
                D_table = f32(D_table + f32(df * df))
                L1_table = f32(L1_table + np.abs(df))
            for tau in (float(np.sort(E)[9]), float(np.sort(E)[0]), float(np.sort(E)[n // 2]), float(np.sort(E)[9]) * 1.5 + (1e-9 if degenerate else 0.0)):
                B = q8_threshold(metric, tau, nqd, q1, float(t_q), R, float(d), float(D_table), nc, float(s_c), radius,
                                 -2.0 if cell_frame == "cs" else (float(L1_table) if cell_frame else -1.0))
                h = Pq @ H
                l = Pq @ L
                assert np.abs(h).max() < 2 ** 22 and np.abs(l).max() < 2 ** 22
                s0 = f32(f32(s_c) * f32(f32(t_q) / f32(254.0)))
                within = E <= f32(tau)
                # plain accumulators: x = fl(fl(fl(254 h + l) * s0 + nhn) - B) >= 0
                fv = fma32(h.astype(f32), f32(254.0), l.astype(f32))
                y = fma32(fv, s0, nhn)
                passed_plain = y >= B
                # biased accumulators: the floats 12582912 + h, 12582912 + l; the threshold moves by 255 * 12582912 * s0
                fvb = fma32((B0 + h).astype(f32), f32(254.0), (B0 + l).astype(f32))
                bb = fma32(f32(K255), s0, B)
                passed_bias = fma32(fvb, s0, nhn) >= bb
                assert not np.any(within & ~passed_plain), (case, metric, qi, tau, int(np.sum(within & ~passed_plain)))
                assert not np.any(within & ~passed_bias), (case, metric, qi, tau, int(np.sum(within & ~passed_bias)))
                # the first half of a tile's product (quick8): max h of the tile, lb = 127 max ||P'||_1, the tile's largest norm word
                for t0 in range(0, n, 32):
                    rows = order[t0:t0 + 32]
                    if not within[rows].any():
                        continue
                    lb = f32(127.0 * float(np.abs(Pq[rows]).sum(axis=1).max()))
                    nmax = nhn[rows].max()
                    xub = fma32(fma32(f32(h[rows].max()), f32(254.0), lb), s0, nmax)
                    assert xub >= B, (case, metric, qi, tau, t0)
                    xub_b = fma32(fma32(f32(B0 + h[rows].max()), f32(254.0), f32(lb + f32(B0))), s0, nmax)
                    assert xub_b >= bb, (case, metric, qi, tau, t0)
    # the bound is not vacuous: at the 10th distance it rejects most of the cell on clustered data
    if case in ("plain", "offset"):
        assert passed_plain.mean() < 0.9
