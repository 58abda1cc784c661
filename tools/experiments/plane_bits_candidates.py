#!/usr/bin/env python3
"""How many candidates would a first-level plane of fewer than 8 bits leave?  (review item: "measure whether a 4-bit first-level plane pays for its looser
bound — measure candidates first").  CPU model of one NS mixture component: 97 656 rows ~ N(c, 0.35^2 I) in 128 dimensions (100M rows / 1024 centres), queries from
the same component, tau = the query's exact 10th-neighbour distance (the tightest threshold any pass can have).  Rows are quantised around the centre with
2^(b-1)-1 levels per side as the int8 plane is (csrc/cells.hip: P' = round((p - c)/s), s = max|p - c| / levels), the query is NOT quantised (the product's H + L/254
split is exact to 2^-16), and a row is a candidate iff its bound can reach tau with the product's own error term |E| <= (s/2) |q - c|_1.  Prints the
candidate fraction per bit width and what that means in bytes per batch: plane bytes saved against re-scored rows (512 B each) added."""
import numpy as np

def main():
    rng = np.random.default_rng(7)
    d, n, nq, k, sigma = 128, 97656, 256, 10, 0.35
    P = (rng.standard_normal((n, d)) * sigma).astype(np.float32)          # residuals p - c
    Q = (rng.standard_normal((nq, d)) * sigma).astype(np.float32)         # queries of the same component, centred
    pn = (P.astype(np.float64) ** 2).sum(1)
    G = Q.astype(np.float64) @ P.astype(np.float64).T                      # q.p
    D = (Q.astype(np.float64) ** 2).sum(1)[:, None] - 2 * G + pn[None, :]
    tau = np.partition(D, k - 1, axis=1)[:, k - 1]
    l1 = np.abs(Q.astype(np.float64)).sum(1)
    amax = np.abs(P).max()
    print(f"component: {n} rows, d {d}, sigma {sigma}; mean tau {tau.mean():.2f}, mean |q-p|^2 {D.mean():.2f}, max|p-c| {amax:.3f}")
    rows_rest, batch_tasks = 89.2e6, 10_000
    for bits in (8, 6, 5, 4):
        levels = 2 ** (bits - 1) - 1
        s = amax / levels
        Pq = np.rint(P.astype(np.float64) / s) * s
        e = P.astype(np.float64) - Pq                                      # the row's own quantisation error, known at build: q.e is NOT known
        # the plane's norm word keeps -|p|^2/2 exactly; unknown at query time is q.e, bounded by (s/2)|q|_1
        Gq = Q.astype(np.float64) @ Pq.T
        lower = (Q.astype(np.float64) ** 2).sum(1)[:, None] - 2 * (Gq + (s / 2) * l1[:, None]) + pn[None, :]   # smallest distance the row may have
        frac = float((lower <= tau[:, None]).mean())
        # per batch: 4.27e9 (task, row) pairs are tested at NS; the 8-bit model's fraction is the yardstick for the others
        print(f"bits {bits}: step {s:.4f}, 2E = {(s * l1).mean():.2f} (mean), candidate fraction {frac:.2e}")
        yield bits, frac

if __name__ == "__main__":
    res = dict(main())
    pairs, cand8 = 4.27e9, 3.08e6   # measured at NS (bench line: filter_pairs, filter_candidates)
    for bits, frac in res.items():
        cand = cand8 * frac / res[8]
        plane = 89.2e6 * (128 * bits / 8 + 4)
        print(f"bits {bits}: rest-pass plane {plane / 1e9:.2f} GB, candidates ~{cand / 1e6:.1f}M = {cand * 512 / 1e9:.2f} GB of rows to re-score -> {(plane + cand * 512) / 1e9:.2f} GB")


def error_terms():
    """Second question (same model, 8 bits): how much tighter than the product's (s/2)|q - c|_1 would the Cauchy-Schwarz term |q - c|_2 * max_row |e|_2 be,
    with the cell's LARGEST row error norm known at build time (not the worst case (s/2) sqrt(d) that GPANN_FILTER_Q8_L1=0 falls back to)?"""
    rng = np.random.default_rng(7)
    d, n, nq, k, sigma = 128, 97656, 256, 10, 0.35
    P = (rng.standard_normal((n, d)) * sigma)
    Q = (rng.standard_normal((nq, d)) * sigma)
    pn = (P ** 2).sum(1)
    G = Q @ P.T
    D = (Q ** 2).sum(1)[:, None] - 2 * G + pn[None, :]
    tau = np.partition(D, k - 1, axis=1)[:, k - 1]
    s = np.abs(P).max() / 127
    Pq = np.rint(P / s) * s
    e = P - Pq
    Gq = Q @ Pq.T
    qn = (Q ** 2).sum(1)
    terms = {"(s/2)|q|_1 (the product)": (s / 2) * np.abs(Q).sum(1),
             "|q|_2 (s/2) sqrt(d) (worst case)": np.sqrt(qn) * (s / 2) * np.sqrt(d),
             "|q|_2 max_row|e|_2 (per cell, known at build)": np.sqrt(qn) * np.sqrt((e ** 2).sum(1).max()),
             "0 (no quantisation error at all)": np.zeros(nq)}
    for name, E in terms.items():
        lower = qn[:, None] - 2 * (Gq + E[:, None]) + pn[None, :]
        if name.startswith("0"):
            lower = D
        print(f"error term {name}: mean 2E = {2 * E.mean():.3f}, candidate fraction {(lower <= tau[:, None]).mean():.2e}")


if __name__ == "__main__":
    error_terms()
