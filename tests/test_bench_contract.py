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
