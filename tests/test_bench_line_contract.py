"""SCHEMA check of committed artefacts (not behavioural coverage: tests/test_gpu_bench_line.py runs bench.py itself on a GPU and validates the line it prints).
The committed bench lines (profiles/r03 … r06/bench_*.json, each the single JSON line `python bench.py ...` printed on an MI355X) carry every field of the
measurement contract: metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
config.workload, the `roofline` object (bound, achieved, peak, unit, frac, traffic) and — where the CPU leg ran — `cpu_baseline`; and the self-checks
of the run (filter mode == exact mode, GPU == CPU oracle) are recorded as true."""
import glob
import json
import os

import pytest

from conftest import ROOT

FILES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03", "bench_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r04", "bench_*.json")) +
               glob.glob(os.path.join(ROOT, "profiles", "r05", "bench_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r06", "bench_*.json")))


def _line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(os.path.dirname(f)) + "/" + os.path.basename(f) for f in FILES])
def test_bench_line_has_the_contract_fields(path):
    j = _line(path)
    if "bench_line" in j:          # (scale_emulated-style wrappers keep the line inside)
        j = j["bench_line"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in j, (os.path.basename(path), key)
    assert j["unit"] == "queries/s" and j["higher_is_better"] is True and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - j["config"]["queries"] / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]      # whole-job QPS = queries / step time
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, (os.path.basename(path), key)
    assert r["bound"] in ("hbm", "mfma") and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    if r["traffic"] is not None:   # PMC bytes of the dominant launch: at least what it has to read, and not much more
        assert 0.98 * r["kernel_bytes"] <= r["traffic"] <= 1.15 * r["kernel_bytes"], (r["traffic"], r["kernel_bytes"])
    ex = j["detail"].get("exact_mode")
    if ex is not None:
        assert ex["results_identical_to_filter_mode"] is True
    c = j.get("cpu_baseline")
    if c:
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in c
        assert c["kind"] in ("port", "reference") and c["results_match_gpu"] is True


@pytest.mark.parametrize("rnd", ["r03", "r04", "r05", "r06"])
def test_the_default_line_is_the_metric_workload(rnd):
    path = os.path.join(ROOT, "profiles", rnd, "bench_ns_default.json")
    if not os.path.exists(path):
        pytest.skip(f"no {rnd} default line committed yet")
    j = _line(path)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert j["metric"].startswith("QPS @ recall@10") and "QPS" in base["metric"]
    cfg = j["config"]
    assert (cfg["n"], cfg["d"], cfg["queries"], cfg["shards"], cfg["k"]) == (100_000_000, 128, 10_000, 40, 10)
    assert cfg["recall_at_k"] >= 0.9 and j["n_gpus"] == 1 and j["cpu_baseline"]["results_match_gpu"] is True
    assert j["hbm_resident_GB"] <= 85.0
    if rnd >= "r05":   # the rest pass shares the chip in the timed loop: both of its times are in the line; matrix-core figures are counted, never above 1
        r = j["roofline"]
        assert r["kernel_ms_alone"] <= r["kernel_ms"] * 1.05 and 0.6 <= r["frac_alone"] <= 1.0 and "overlap_note" in r
        assert 0.0 < r["filter_mfma_frac_i8"] <= 1.0 and 0.0 < r["filter_mfma_frac_bf16"] <= 1.0 and "filter_mfma_frac" not in r
        assert j["detail"]["router_selfcheck_events"] == 0 and j["detail"]["router_verify_failures"] == 0
        assert j["cpu_baseline"]["value_reference_flags_32_threads"] > 0
