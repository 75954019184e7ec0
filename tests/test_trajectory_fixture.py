"""CPU checks of the full-size trajectory fixture (tests/golden/trajectory_kat.json, tests/golden/make_trajectory_golden.py): it starts
where the single-pass fixture of the same configuration stands, its learning rates are the staircase of W/train.py:301-311, and it
crosses a step of that staircase (the HIP replay is tests/test_gpu_trajectory.py)."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import np_ref  # noqa: E402


def test_trajectory_fixture_is_consistent_with_the_single_pass_fixture_and_the_lr_staircase():
    tr = json.load(open(os.path.join(HERE, "golden", "trajectory_kat.json")))
    one = json.load(open(os.path.join(HERE, "golden", "fullsize_kat.json")))[tr["config"]]
    assert tr["steps"][0]["loss"] == pytest.approx(one["loss"], rel=1e-12)          # same weights, same batch, same restatement
    hy = tr["hyper"]
    for s, st in enumerate(tr["steps"]):
        want = np_ref.exponential_decay(hy["base_lr"], s, hy["batch_size"], hy["decay_examples"], hy["decay"])
        assert st["learning_rate"] == pytest.approx(float(want), rel=1e-12)
        assert set(st["grad_norms"]) == set(tr["params"])
    assert len({st["learning_rate"] for st in tr["steps"]}) >= 2
    assert len(tr["steps"]) >= 3 and all(p["n"] > 0 for p in tr["params"].values())
