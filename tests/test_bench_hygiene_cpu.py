"""CPU checks of bench.py's measurement hygiene added in round 6 (ADVICE r05 / VERDICT r05 "next" 9): the wall-clock budget of the extras, the CPU
baseline's protocol (small warm-up, timed passes of the full sample, the memory guard's wording), the draw buffers' initialisation."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_extras_budget_hands_out_what_is_left_and_refuses_below_the_floor():
    b = bench.ExtrasBudget(100.0)
    assert 99.0 < b.timeout(600) <= 100.0 and b.timeout(40) == 40.0
    b.t_end = time.perf_counter() + 20.0
    assert b.timeout(600) is None and 19.0 < b.timeout(600, floor=10.0) <= 20.0


def test_cpu_baseline_protocol_on_a_tiny_sample():
    out = bench.cpu_baseline(64, 16, 48, "fg")
    assert out["kind"] == "port" and out["unit"] == "rays/s" and out["value"] > 0 and out["cores"] >= 1
    assert "2 frames x 48 rays x 16 samples" in out["sample"] and "timed pass" in out["sample"]


def test_distinct_indices_never_returns_uninitialised_slots():
    g = torch.Generator().manual_seed(0)
    x = bench.distinct_indices(5000, 1024, "cpu", g)
    assert x.shape == (1024,) and int(x.min()) >= 0 and int(x.max()) < 5000 and x.unique().numel() == 1024
    # an adversarial case: almost every draw a duplicate -> the tail repeats a valid index instead of holding garbage
    class G:  # a generator stand-in is not needed: patch randint
        pass
    real = torch.randint
    try:
        torch.randint = lambda lo, hi, size, **kw: torch.zeros(size, dtype=torch.int64)
        y = bench.distinct_indices(100000, 16, "cpu", None)
    finally:
        torch.randint = real
    assert int(y.min()) == 0 and int(y.max()) == 0
