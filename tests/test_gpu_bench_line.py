"""GPU: bench.py end to end on a small workload — the line it PRINTS (not a committed artefact) carries the measurement contract, its self-checks ran
(filter mode == exact mode, GPU == CPU oracle on the sampled queries), the roofline block is consistent with the run's own kernel time and the
traffic lookup refuses a PMC pass of another workload."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_prints_a_line_that_meets_the_contract(tmp_path):
    env = dict(os.environ, GPANN_BENCH_CPU_QUERIES="200")
    env.pop("GPANN_ROUTER_VERIFY", None)      # (the timed configuration: the shipping path of the router, deferred scan calls)
    env.pop("GPANN_ROUTER_STRICT", None)
    env.pop("GPANN_FILTER_SMALL_SHARD_ROWS", None)   # (... and the shipped pass structure for a store of this size)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--points", "400000", "--d", "64", "--shards", "6", "--nq", "1000", "--steps", "4", "--warmup", "2",
           "--router-budget", "3000", "--search-budget", "50000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "latency_ms_one_batch"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["warmup"] == 2 and j["unit"] == "queries/s" and j["higher_is_better"] is True
    assert j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic" and "model" not in j["config"]
    assert abs(j["value"] - 1000 / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    assert j["config"]["recall_at_k"] >= 0.9 or j["config"]["nprobe"] == 6
    ro = j["roofline"]
    assert ro["bound"] == "hbm" and ro["peak"] == 8000.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9
    assert abs(ro["achieved"] - ro["kernel_bytes"] / (ro["kernel_ms"] * 1e-3) / 1e9) <= 1e-6 * max(ro["achieved"], 1.0)   # achieved = this run's bytes / this run's kernel time
    assert ro["traffic"] is None                       # no committed PMC pass describes this workload: the lookup must refuse, not borrow one
    assert 0.0 < ro["exact_mode_frac"] < 1.0
    assert j["detail"]["exact_mode"]["results_identical_to_filter_mode"] is True
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["results_match_gpu"] is True and c["cores"] >= 1 and c["value"] > 0
    assert c["sample_rows_per_query"] > 0 and c["batch_rows_per_query"] > 0 and c["value_scaled_to_batch"] > 0
    assert "hard_workload" not in j["detail"]          # (only the default workload carries that leg)
    assert "deferred" in j["config"]["pipeline"] or "enqueued behind" in j["config"]["pipeline"]


def test_bench_starts_its_own_ranks_without_a_launcher(tmp_path):
    """`python3 bench.py --gpus 2` with no WORLD_SIZE in the environment — how the driver starts it — must spawn the two ranks itself
    (bench.py spawn_ranks) and print rank 0's ONE line; here both ranks share the one GPU over the TCP backend."""
    env = dict(os.environ, GPANN_BENCH_FORCE_DEVICE="0", GPANN_COMM_WAIT_S="120")
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GPANN_COMM_PORT", "GPANN_ROUTER_VERIFY", "GPANN_ROUTER_STRICT", "GPANN_FILTER_SMALL_SHARD_ROWS"):
        env.pop(name, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "tcp", "--points", "400000", "--d", "64", "--shards", "6", "--nq", "1000",
           "--steps", "3", "--warmup", "1", "--router-budget", "3000", "--no-cpu-baseline", "--no-exact-compare"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["world_size_seen"] == 2
    assert j["scaling"] == "strong" and j["value"] > 0
    assert "launcher: started 2 ranks" in r.stderr


def test_matrix_core_figures_are_counted_not_modelled(tmp_path):
    """roofline.filter_mfma_*: issued matrix instructions from the kernels' own counters, each dtype against its own peak — a cloud without cluster
    structure (every query tile on every row tile: where the old pairs x d x 6 / bf16-peak figure read 2.3) must give fractions in (0, 1]"""
    env = dict(os.environ)
    env.pop("GPANN_ROUTER_VERIFY", None)
    env.pop("GPANN_ROUTER_STRICT", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--points", "600000", "--d", "128", "--shards", "6", "--nq", "1000", "--steps", "3", "--warmup", "1",
           "--router-budget", "3000", "--dist", "uniform", "--nprobe", "3", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.split("\n") if ln.strip()][-1])
    ro = j["roofline"]
    assert "filter_mfma_frac" not in ro and "filter_mfma_TFLOPs" not in ro
    issued = ro["filter_mfma_issued"]
    assert issued["i8_32x32x32"] > 0 and issued["bf16_32x32x16"] > 0 and 0 < issued["i8_32x32x32_rest_pass"] <= issued["i8_32x32x32"]
    assert 0.0 < ro["filter_mfma_frac_i8"] <= 1.0 and 0.0 < ro["filter_mfma_frac_bf16"] <= 1.0 and 0.0 < ro["rest_pass_mfma_frac_i8"] <= 1.0
    assert ro["filter_mfma_frac_i8"] + ro["filter_mfma_frac_bf16"] <= 1.0
    # every (row tile, query tile) product of an 8-bit pass issues KS = d / 32 instructions for H.P' and at most as many again for L.P'
    assert j["detail"]["router_selfcheck_events"] == 0 and j["detail"]["router_verify_failures"] == 0
    assert j["detail"]["exact_mode"]["results_identical_to_filter_mode"] is True
