"""RescoreKernel's arithmetic (gp-ann_amd/csrc/scan_filter.h), restated in numpy float32 and checked against the strict oracle on the CPU: the kernel
lets the lane that LOADED a group of four dimensions compute the group's term — ((d0 d0 + d1 d1) + d2 d2) + d3 d3, or the four products of the inner
product — leaves the terms in LDS and has one lane add a row's terms in the reference's order (src/dist.cpp:17-52: `result += <group term>`, then the
tail dimensions one by one).  The claim the kernel rests on: the terms do not depend on `result`, so computing them elsewhere first changes no bit."""
import numpy as np
import pytest

f32 = np.float32


def rescore_l2(p, q):
    d = p.shape[0]
    full, tail = d // 4, d % 4
    terms = []
    for g in range(full):          # phase A: any lane, any time
        d0, d1, d2, d3 = (f32(p[4 * g + i]) - f32(q[4 * g + i]) for i in range(4))
        t = f32(f32(d0 * d0) + f32(d1 * d1))
        t = f32(t + f32(d2 * d2))
        terms.append(f32(t + f32(d3 * d3)))
    for i in range(full * 4, full * 4 + tail):
        di = f32(p[i]) - f32(q[i])
        terms.append(f32(di * di))
    result = f32(0)
    for t in terms:                 # phase B: one lane, the reference's order
        result = f32(result + t)
    return result


def rescore_ip(p, q):
    result = f32(0)
    for prod in [f32(f32(q[i]) * f32(p[i])) for i in range(p.shape[0])]:
        result = f32(result + prod)
    return f32(f32(1) - result)


@pytest.mark.parametrize("d", [1, 3, 4, 7, 64, 96, 100, 128, 200, 255, 256])
def test_terms_first_then_the_references_order_is_the_references_distance(orc, d):
    rng = np.random.default_rng(1000 + d)
    q = (rng.standard_normal(d) * 3).astype(np.float32)
    pts = (rng.standard_normal((40, d)) * 3 + rng.standard_normal(d)).astype(np.float32)
    pts[0] = q                         # distance zero
    pts[1] = q + np.float32(1e-4)      # cancellation
    want_l2 = orc.distance_many(pts, q, 0)
    want_ip = orc.distance_many(pts, q, 1)
    for r in range(pts.shape[0]):
        assert rescore_l2(pts[r], q).view(np.uint32) == want_l2[r].view(np.uint32), (d, r)
        assert rescore_ip(pts[r], q).view(np.uint32) == want_ip[r].view(np.uint32), (d, r)
