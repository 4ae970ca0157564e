"""bench.py's last stdout line stays small enough for the driver to parse (a 20.8 KB line left
BENCH_r05.parsed = null).  Stand-in numbers, the real builder."""
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _full_record(cfg="c2"):
    long = "x" * 1500
    legs = {k: {"evals_per_s": 1234.56789012345, "evals": 1000, "seconds": 2.7123456789, "threads": 16, "semantics": long}
            for k in ("one_core_in_transit", "one_core_every_cadence", "all_cores_in_transit", "all_cores_every_cadence",
                      "c3_one_core", "c3_all_cores", "c4_one_core", "c4_all_cores", "c5_one_core", "c5_all_cores")}
    roof = {"bound": "hbm", "kernel": long, "achieved": 5025.794356979103, "peak": 8000.0, "unit": "GB/s",
            "frac": 0.6282242946223879, "frac_definition": long, "traffic": 1468878319.8452308, "traffic_source": long,
            "algorithmic_bytes_per_launch": 1423807536, "algorithmic_bytes_breakdown": {"a": 1, "b": 2},
            "active_cadences_per_launch": 4062657, "kernel_ms": 0.28329999893903735,
            "kernel_ms_quantiles": {"iters": 20, "median_ms": 0.28}, "frac_step": 0.6058795658880853, "frac_step_definition": long,
            "survey_8d_count": {"bytes_per_unit": 24, "bytes_per_launch": 3686400000, "GBps": 13012.3, "note": long},
            "pmc": {"dominant_kernel_traffic_bytes": 1.4e9, "dominant_kernel_rocprof_avg_us": 262.2, "dominant_kernel_GBps": 5590.1,
                    "dominant_kernel_frac": 0.6987, "definition": long},
            "valu": {k: {"insts": 1e9, "frac": 0.6, "note": long} for k in "abcdefgh"}}
    return {
        "metric": "light-curve evals/sec (value+grad) at 150k cadences", "value": 3485980.5684826737, "unit": "evals/s",
        "n_gpus": 1, "steps": 20, "warmup": 5, "setup_steps_before_warmup": 300, "value_after_contract_warmup_only": 3.4e6,
        "ms_per_step": 0.2937480516266078, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": long, "n_cadences": 150000, "draws_per_gpu": 1024, "global_draws": 1024,
                   "setup_steps_before_warmup": 300, "parallelism": long, "step": long},
        "timing": {"iters": 6000, "median_ms": 0.29, "p10_ms": 0.28, "p90_ms": 0.30, "mean_ms": 0.29},
        "roofline": roof,
        "extras": {f"leg{i}": {"median_ms": 1.0, "note": long} for i in range(25)},
        "cpu_baseline": {"value": 864.123456789, "unit": "evals/s", "cores": 1, "kind": "port",
                         "leg": {"c2": "one_core_in_transit"}.get(cfg, cfg + "_one_core"), "sample": long,
                         "cpu_model": "AMD EPYC 9575F 64-Core Processor", "host_cores": 256, "usable_cores": 16,
                         "usable_cores_from": long, "legs": legs, "note": long},
        "configs_ms": {"c2": 0.2897, "c3": 3.3668, "c4_64": 0.116, "c5_128": 1.8462, "c5b": 2.4611, "sparse": 0.1822,
                       "chi2": 0.1877, "hmc": 1.7506, "nuts_leaf": 0.2159, "c2_one_call": 0.29, "kepler_GBps": 5000.0,
                       "c5_j10": 9.1, "c5_kappa1e9": 3.0, "note": long},
        "full_record": "gpurun_out/bench_full.json",
    }


@pytest.mark.parametrize("cfg", ["c2", "c3", "c5"])
def test_compact_line_is_small_and_complete(cfg):
    import bench

    full = _full_record(cfg)
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.COMPACT_LIMIT <= 4096, len(text)
    assert "\n" not in text
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "configs_ms"):
        assert k in back, k
    assert back["value"] == pytest.approx(full["value"], rel=1e-7)
    assert back["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-7)
    assert "model" not in back["config"] and "workload" in back["config"]
    assert all(not isinstance(v, str) or len(v) <= 200 for v in back["config"].values())
    r = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    c = back["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["all_cores_value"] == pytest.approx(1234.57, rel=1e-4) and c["all_cores_threads"] == 16
    assert "legs" not in c and "extras" not in back
    assert back["configs_ms"]["c5_128"] == 1.8462


def test_compact_line_without_optional_blocks():
    import bench

    full = _full_record()
    full["cpu_baseline"] = None
    full["roofline"]["pmc"] = None
    full["roofline"]["traffic"] = None
    line = bench.compact_line(full)
    assert line["cpu_baseline"] is None and line["roofline"]["traffic"] is None
    assert len(json.dumps(line)) < bench.COMPACT_LIMIT
