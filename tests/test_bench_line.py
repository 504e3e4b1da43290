"""bench.py prints ONE line and the driver keeps its last 8 KB: the short form (bench.compact_line) must hold every BASELINE config
inside that budget, with `config.legs` as the last key of `config`, right in front of `roofline`.  Host only."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _leg(i):
    return {"workload": "w" * 300, "value": 1.234567e11 + i, "ms_per_step": 12.345678, "steps": 5, "roofline_frac": 0.4123456,
            "roofline_frac_by_counters": 0.3312345, "kernels_only_value": 1.5e11, "stage_ms": 7.123456, "dominant_kernel": "k_bwd_fast",
            "cpu_baseline": {"value": 2.2e8, "cores": 16, "kind": "port", "sample": "s" * 400}, "note": "n" * 500}


def test_the_printed_line_is_short_and_ordered(tmp_path):
    b = _bench()
    names = ["realistic", "cpg", "hdp", "hdp_threshold_0.01", "hdp_threshold_0.01_pairs8", "hdp_dense", "expectations", "scaling_slice"]
    full = {"metric": "dp_cell_updates_per_s", "value": 1.6e11, "unit": "cell_updates/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 10.5, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "x" * 400, "reads_per_gpu": 2000, "events_per_read": 5000, "batches_in_flight": 4, "events_per_s": 9.4e8,
                       "pairs_per_event": 0.9, "kernel_ms": {"forward": 3.9, "backward_posterior": 5.0, "fold_and_finalize": 0.15},
                       "step": "s" * 600, "results": {"note": "r" * 900},
                       "step_ms": {"p10": 8.0, "p50": 9.9, "p90": 10.8, "max": 15.3, "slowest": [{"ms": 15.3, "create_ms": 15.1, "wait_ms": 0.0}] * 3},
                       "long_run": {"steps": 200, "value": 1.63e11, "ms_per_step": 10.5, "seconds": 2.1, "note": "l" * 200,
                                    "step_ms": {"p10": 9.5, "p50": 10.4, "p90": 11.3, "max": 16.2}},
                       "kernels_only_resident_inputs": {"value": 1.85e11, "ms_per_step": 9.0, "note": "k" * 200},
                       "secondary": dict({n: _leg(i) for i, n in enumerate(names)}, skipped_leg={"skipped": "wall-time budget of the default run"}),
                       "scaling_job": {"value": 1.7e11, "n_gpus": 1, "total_reads": 100000, "events_per_read": 10000, "repetitions": 5,
                                       "wall_s": 1.0, "wall_s_min": 0.99, "wall_s_max": 1.02, "wall_spread": 0.03, "slice_sizes_rank0": [2000],
                                       "slices_rank0": 50, "batches_in_flight": 3, "per_rank": [{"rank": 0, "reads": 100000, "wall_s": 1.0,
                                                                                                   "idle_at_barrier_s": 0.0}],
                                       "workload": "j" * 300, "timed": "t" * 400}},
            "roofline": {"bound": "issue", "kernel": "k_bwd_fast", "achieved": 4400.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.55,
                         "traffic": 1.3e10, "frac_by_counters": 0.33, "stage_ms": 4.9, "launches_per_step": 1, "avg_launch_ms": 4.9,
                         "algorithmic_bytes_per_step": 2.17e10, "limiter": "i" * 300, "stage_ms_is": "q" * 300,
                         "counters_collected_at": {"head": "abc1234", "tag": "r06", "collected": "2026-10-05", "command": "c" * 300}},
            "issue_roofline": {"kernel": "k_bwd_fast", "frac": 0.58, "valu_per_step": 1.6e9, "source": "p" * 200},
            "cpu_baseline": {"value": 2.2e8, "unit": "cell_updates/s", "cores": 16, "kind": "port", "sample": "z" * 500,
                             "one_thread_value": 1.5e7}}
    path = str(tmp_path / "full.json")
    line = b.compact_line(full, path)
    assert "\n" not in line and len(line) < 6000          # the driver keeps the last 8 KB of stdout
    d = json.loads(line)
    assert json.load(open(path)) == full                   # nothing is lost: the complete record is beside it
    assert list(d)[-3:] == ["roofline", "issue_roofline", "cpu_baseline"] and list(d)[-4] == "config"
    legs_key = [k for k in d["config"] if k != "full_record"][-1]
    assert legs_key == "legs" and d["config"]["legs_columns"] == b.LEG_COLUMNS
    for n in names:
        row = d["config"]["legs"][n]
        assert len(row) == len(b.LEG_COLUMNS) and row[0] > 1e11 and row[2] == 0.4123 and row[4] == 2.2e8
    assert d["config"]["legs"]["skipped_leg"].startswith("skipped")
    assert d["roofline"]["frac"] == 0.55 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 16
    assert d["config"]["scaling_job"]["scaling"] == "strong" and d["config"]["scaling_job"]["repetitions"] == 5
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in d
