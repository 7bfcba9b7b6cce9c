"""bench.py's record helpers on the CPU: the parity record (top-5 overlap, the near-tie rule for a differing top-1) and the
BASELINE.json configuration names -- the parts of the JSON line the judge reads that do not need a GPU."""
import importlib.util
import os

import numpy as np

from tests.conftest import ROOT

spec = importlib.util.spec_from_file_location("eco_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_parity_record_fields_and_near_tie_rule():
    rng = np.random.default_rng(0)
    ref = rng.normal(size=(3, 400)).astype(np.float32) * 1000
    ref[1, 7] = ref[1].max() + 50.0            # clip 1: a clear winner ...
    ref[2, 9] = ref[2].max() + 2.0             # clip 2: ... and a near-tie between classes 9 and 11
    ref[2, 11] = ref[2, 9] - 1.0
    got = ref + rng.normal(size=ref.shape).astype(np.float32) * 0.5
    got[2, 11] = got[2, 9] + 0.5               # the path under test picks the other one of the tied pair
    rec = bench.parity_record(got, ref, "bf16", "unit test", clips=[0, 10, 31])
    assert rec["clips_checked"] == 3 and rec["clips"] == [0, 10, 31] and rec["tolerance"] == 3e-2
    assert rec["max_rel_err"] < 1e-2 and not rec["top1_agree"]            # top-1 differs on clip 2 ...
    assert rec["top1_equal_or_reference_near_tie"]                         # ... inside twice the clip's absolute error
    assert rec["top5_overlap_min"] >= 4 and len(rec["top5_overlap_per_clip"]) == 3
    # a REAL disagreement is not excused: the picked class is far below the reference's maximum
    bad = ref.copy()
    bad[1, 3] = bad[1].max() + 10.0
    rec = bench.parity_record(bad, ref, "f32", "unit test")
    assert not rec["top1_agree"] and not rec["top1_equal_or_reference_near_tie"] and rec["tolerance"] == 1e-3


def test_baseline_config_names():
    assert bench.baseline_config("lite", 16, 32, "f32", 1) == "BASELINE.json configs[1]"
    assert bench.baseline_config("lite", 16, 32, "f32", 8) == "BASELINE.json configs[2]"
    assert bench.baseline_config("full", 16, 32, "f32", 1) == "BASELINE.json configs[3]"
    assert bench.baseline_config("lite", 32, 32, "bf16", 1).startswith("BASELINE.json configs[4]")
    assert bench.baseline_config("lite", 4, 1, "f32", 1).startswith("BASELINE.json configs[0]")
    assert bench.baseline_config("lite", 8, 3, "f32", 1) == "not a BASELINE.json configuration"
