"""bench.py's `roofline.traffic` is looked up in the committed PMC summaries (profiles/*/*_pmc_traffic.json) by kernel-name fragments:
a change of the kernel's template arguments must not silently send the lookup back to an older round's file."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_traffic_of_the_ns_rest_pass_comes_from_the_newest_round():
    b = _bench()
    traffic, src = b.lookup_traffic(100_000_000, 128, 40, 1, "FilterKernelCells<", {"": 1}, largest_launch=True)
    assert src is not None and src.startswith("profiles/r06/"), src      # (the newest committed pass of this workload)
    # the rest pass reads about one int8 plane + norm word per row of the store: 100M x 132 B, give or take the pruned / repeated cells
    assert 10e9 < traffic < 15e9, traffic
    # and it is only reported for a run whose rest pass has to read the same bytes
    assert b.lookup_traffic(100_000_000, 128, 40, 1, "FilterKernelCells<", {"": 1}, largest_launch=True, kernel_bytes=11.78e9)[0] == traffic
    assert b.lookup_traffic(100_000_000, 128, 40, 1, "FilterKernelCells<", {"": 1}, largest_launch=True, kernel_bytes=23.9e9)[0] is None


def test_traffic_lookup_knows_the_other_configurations_and_nothing_else():
    b = _bench()
    for shape in [(1_000_000, 128, 8, 1), (10_000_000, 96, 16, 4), (100_000_000, 200, 40, 1), (200_000_000, 100, 60, 1)]:
        traffic, src = b.lookup_traffic(*shape, "FilterKernelCells<", {"": 1}, largest_launch=True)
        assert src is not None and traffic > 0, shape
    assert b.lookup_traffic(123, 7, 3, 1, "FilterKernelCells<", {"": 1}, largest_launch=True) == (None, None)


def test_a_pass_taken_on_other_bytes_is_not_reported():
    """with the run's own dominant-launch bytes given, a committed pass is only used if ITS run read the same bytes (within 2 %): the
    round-2 files carry no such figure (taken before the guard existed, on a layout with different bytes per row) and must not be used"""
    b = _bench()
    traffic, src = b.lookup_traffic(10_000_000, 96, 16, 4, "FilterKernelCells<", {"": 1}, largest_launch=True, kernel_bytes=15.9e9)
    assert traffic is None or src.startswith("profiles/r04/"), (traffic, src)
    if traffic is None:
        assert src is None or src.startswith("stale: "), src
