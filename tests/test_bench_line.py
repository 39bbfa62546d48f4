"""The driver keeps a few KB of bench.py's stdout: the ONE JSON line must stay small whatever the legs hold (round 5's 23 KB line was
truncated and never parsed).  Builds the digest from a canned full object (round 5's own, plus --gpus 8-shaped blocks)."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _canned():
    with open(os.path.join(ROOT, "profiles", "r05_bench_output.json")) as f:
        out = json.load(f)
    out["n_gpus"] = 8
    out["scaling"] = "weak"
    out["config"]["parallelism"] = "lists % 8, ROUTED: " + "x" * 400
    out["multi_gpu"] = {"mode": "routed",
                        "routed": {"queries_per_step_per_rank": 4096, "routed_pairs_per_step_by_rank": [4096.0 + r for r in range(8)],
                                   "qps": 31234567.8, "stage_ms_rank0": {f: 0.1234 for f in ("shard_exchange", "coarse_pass", "ivf_plan", "ivf_scan",
                                                                                            "ivf_sample_scan", "rerank", "merge")},
                                   "note": "n" * 300},
                        "replicated": {"qps": 9876543.2, "ms_per_step": 0.4147, "note": "m" * 300}}
    out["c4_sharded"] = {"workload": "w" * 400, "rows_on_rank0": 12499000, "build_s": 9.9, "scaling": "weak",
                         "batches": {"4096": {"qps": 601234.5, "ms_per_batch": 6.81}, "1024": {"qps": 191234.5, "ms_per_batch": 5.35}}}
    return out


def test_line_is_small_and_round_trips():
    import bench

    out = _canned()
    line = bench.compact_line(out)
    assert "\n" not in line and len(line) < bench.LINE_BUDGET < 8000
    got = json.loads(line)
    for key in CONTRACT:
        assert key in got, key
    assert got["value"] == out["value"] and got["ms_per_step"] == out["ms_per_step"] and got["n_gpus"] == 8
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in got["roofline"], key
    assert got["roofline"]["frac"] == out["roofline"]["frac"] and "note" not in got["roofline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in got["cpu_baseline"], key
    assert "workload" in got["config"] and "model" not in got["config"]
    legs = got["legs"]
    assert legs["multi_gpu"]["routed"]["routed_pairs_per_step_by_rank"] == out["multi_gpu"]["routed"]["routed_pairs_per_step_by_rank"]
    assert legs["c4_sharded"]["batches"]["4096"]["qps"] == 601234.5
    assert legs["target_100m"]["batches"]["4096"]["roofline_frac"] == out["target_100m"]["batches"]["4096"]["roofline_frac"]
    assert legs["C5"]["bm25_batch1024"]["hbm_frac"] == out["other_configs"]["C5"]["bm25_batch1024"]["hbm_frac"]
    assert legs["iid"]["exhaustive_flat"]["whole_step_mfma_frac"] == out["iid"]["exhaustive_flat"]["whole_step_mfma_frac"]
    assert legs["latency"]["p50_us"] == out["latency"]["p50_us"]


def test_line_sheds_legs_rather_than_growing():
    import bench

    out = _canned()
    # a leg that explodes (hundreds of batch sizes) must not cost the line
    out["target_100m"]["batches"] = {str(b): {"qps": 1.0 * b, "ms_per_batch": 0.5, "roofline_frac": 0.5, "whole_step_frac": 0.4} for b in range(400)}
    out["multi_gpu"]["routed"]["routed_pairs_per_step_by_rank"] = [4096.0] * 32
    line = bench.compact_line(out)
    assert len(line) < bench.LINE_BUDGET
    got = json.loads(line)
    assert got["value"] == out["value"] and got["roofline"]["frac"] == out["roofline"]["frac"] and got["cpu_baseline"]["value"]
    assert got["legs"]["target_100m"] == "see bench_detail.json"


def test_line_survives_failed_and_missing_legs():
    import bench

    out = _canned()
    for key in ("latency", "iid", "mid", "target_100m", "other_batches", "multi_gpu", "c4_sharded", "concurrent_batches"):
        out[key] = None
    out["other_configs"] = {"C3": {"error": "RuntimeError('x')"}, "C5": {"error": "e" * 500}}
    out["blobs03"] = {"error": "boom"}
    got = json.loads(bench.compact_line(copy.deepcopy(out)))
    assert got["legs"]["C3"]["error"].startswith("RuntimeError") and len(got["legs"]["C5"]["error"]) <= 80
    assert got["legs"]["blobs03"]["error"] == "boom"


def test_emit_writes_detail_and_prints_one_line(tmp_path, capsys, monkeypatch):
    import bench

    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.mkdir(tmp_path / "gpurun_out")
    out = _canned()
    bench.emit(out)
    cap = capsys.readouterr()
    lines = [ln for ln in cap.out.split("\n") if ln]
    assert len(lines) == 1 and json.loads(lines[0])["value"] == out["value"]
    for p in (tmp_path / "bench_detail.json", tmp_path / "gpurun_out" / "bench_detail.json"):
        assert json.loads(p.read_text()) == out
