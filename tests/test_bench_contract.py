"""bench.py's contract, as far as it can be checked without a GPU: flags and defaults the driver relies on, the workloads
against BASELINE.json's configs, and the algorithmic-bytes formula against SURVEY 8(d)'s figures."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _bench():
    import importlib
    return importlib.import_module("bench")


def test_driver_flags_and_defaults(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup, a.mode, a.workload) == (1, 1000, 20, "test", "c2")   # no flags: N = 1, minutes at most
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "50", "--warmup", "5"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 50, 5)
    assert a.graph == "auto" and a.split == "pairs" and a.streams == 1


def test_workloads_are_baseline_configs():
    bench = _bench()
    cfgs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    for key, w in bench.WORKLOADS.items():
        text = cfgs[w["baseline_config"]]
        nums = {int(x) for x in re.findall(r"\d+", text.replace(",", ""))}
        assert w["points"] in nums and w["hyps"] in nums, (key, text)
    w = bench.WORKLOADS
    assert w["c1"]["solver"] == "f8" and w["c1"]["sampler"] == "uniform"
    assert w["c2"]["solver"] == "nister" and w["c3"]["solver"] == "stewenius" and w["c3"]["pairs"] == 32
    assert w["c4"]["solver"] == "rigid"
    assert "2000 pts" in json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]


def test_algorithmic_bytes_match_the_survey():
    bench = _bench()
    # K4 per pair = 16 N + 36 M + 4 M + M N ; C2 (N = 2000, M = 10 240) = 20.92 MB ; C3 = 32 x (81.92 + 1.67) MB
    assert bench.k4_bytes(1, 2000, 10240) == 16 * 2000 + 40 * 10240 + 10240 * 2000
    assert abs(bench.k4_bytes(1, 2000, 10240) / 1e6 - 20.92) < 0.01
    assert abs(bench.k4_bytes(32, 2000, 40960) / 1e9 - 2.675) < 0.002
    # K4r (C4): 24 N + 48 M + 4 M + 4 + M N = 103.7 MB
    assert abs(bench.k4r_bytes(1, 50000, 2048) / 1e6 - 103.7) < 0.1
    assert bench.HBM_PEAK_GBS == 8000.0


def test_committed_bench_line_carries_the_contract_fields_and_the_round6_records():
    """the line the end-of-round script stored (profiles/r6_bench_line.json = `python bench.py` on the GPU box): every field the
    driver's contract names, the roofline / cpu_baseline objects, and the sub-records the reviews of rounds 4 and 5 asked for"""
    d = json.load(open(os.path.join(ROOT, "profiles", "r6_bench_line.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "hypotheses/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["binding_unit"] == "valu_f32"
    # achieved = algorithmic bytes per launch / the launch's HIP-event duration; the duration fits inside the step
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-3 * r["achieved"]
    assert r["avg_launch_ms"] < d["ms_per_step"]
    # the traffic figure is a builder-run PMC capture (hash-checked against the kernel sources), and the line says so
    assert r["traffic"] is not None and 0.95 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.15
    assert "NOT measured in this run" in r["traffic_source"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["single_thread_value"] > 0
    # round 6: the best MEDIAN over 1 / 8 / 32 threads, with its thread count and spread; never below the single-thread figure
    assert c["threads"] == c["cores"] and set(c["by_threads"]) <= {"1", "8", "32"}
    assert abs(c["value"] - max(c["by_threads"].values())) < 0.1        # (the per-thread-count figures are rounded to 0.1)
    assert c["value"] >= c["single_thread_value"] and c["spread"] >= 0
    cfg = d["configs"]
    for k in ("c1", "c3", "c4", "c2_p32", "c2_p1", "c5_train_p32", "dropin_layer_loop"):
        assert k in cfg, k
    for k in ("c1", "c3", "c4", "c2_p32", "c2_p1", "c5_train_p32"):
        assert cfg[k]["ms_per_step"] > 0, k
    for k in ("c1", "c3", "c4", "c5_train_p32"):       # review item 5: a measured CPU baseline next to every config
        b = cfg[k]["cpu_baseline"]
        assert b["kind"] == "port" and b["value"] > 0 and b["single_thread_value"] > 0 and b["cores"] >= 1, k
        assert b["value"] >= b["single_thread_value"] and "by_threads" in b and "spread" in b, k
    assert cfg["c2_p1"]["ms_per_step"] <= 0.12                      # the round-3 review's bar for the one-pair call (replayed)
    # round 5: the train step's MatchLoss is one pass (value + unscaled gradient), sampler + gather one launch each way
    launches = set(cfg["c5_train_p32"]["launch_ms"])
    assert {"dr_match_loss_fused_f32", "dr_gumbel_topk_gather_soft_f32", "dr_gumbel_topk_gather_bwd_f32",
            "dr_solve_nister5_bwd_sel_f32"} <= launches
    assert not ({"dr_episym_fwd_f32", "dr_episym_bwd_mean_f32", "dr_gather_fwd_f32", "dr_gather_bwd_f32"} & launches)
    loop = cfg["dropin_layer_loop"]
    assert loop["test_mode"]["ms_per_pair"] < loop["test_mode_eager"]["ms_per_pair"] and loop["batched_forward_ms_per_pair"] > 0
    # round 6 (review item 2): the reference's per-pair call <= 0.15 ms at -rbs 1024, <= 0.30 ms at the reference's default -rbs 64
    # (two device rounds in one replayed graph), and the wall-time semantics of the layer's second return value beside it
    assert loop["test_mode"]["ms_per_pair"] <= 0.15 and loop["dropin_layer_loop_rbs64"]["ms_per_pair"] <= 0.30
    assert loop["dropin_layer_loop_rbs64"]["device_rounds"] == 2 and loop["test_mode_sync_timing"]["ms_per_pair"] > 0
    # c4: the residual launch timed back to back beside the single-launch figure (review item 3: 43.2 vs 49.7 us)
    b2b = cfg["c4"]["scoring_roofline"]["back_to_back"]
    assert b2b["launches"] == 50 and 0 < b2b["avg_launch_ms"] <= cfg["c4"]["scoring_roofline"]["avg_launch_ms"] * 1.05
    f = d["fused_driver"]["scoring_roofline"]
    assert f["bytes_formula"] == "P (16 N + 40 M + N)" and f["algorithmic_bytes_per_launch"] == 128 * (16 * 2000 + 40 * 10240 + 2000)
    a = d["k4_all_valid"]
    assert a["all_slots_valid"]["evaluated_slot_fraction"] == 1.0 and a["all_slots_valid"]["hbm_frac"] < r["frac"]
