"""Pins oracle/metrics.py (and the product's eval_util) to known answers produced by the IMPORTED reference
metric code (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import metrics as om
from oracle import np_ref
import yt8m_amd.eval_util as eu
import yt8m_amd.average_precision_calculator as apc
import yt8m_amd.mean_average_precision_calculator as mapc


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "metrics_kat.json")) as fh:
        return json.load(fh)


# GAP tolerance: the golden values were produced by the reference code running under numpy 2.2, where NEP-50
# promotion makes its `ap += poscount / (i + 1) * delta_recall` accumulate in float32 (total_num_positives is a
# float32 numpy scalar).  Under the reference's pinned numpy 1.12 the same line accumulates in float64, which is
# what oracle and product do, so agreement is to float32 rounding (a few 1e-8 absolute), not to 1e-12.
GAP_TOL = dict(rel=2e-6, abs=1e-9)

IMPLS = [("oracle", om.gap, om.hit_at_one, om.precision_at_equal_recall_rate),
         ("product", eu.calculate_gap, eu.calculate_hit_at_one, eu.calculate_precision_at_equal_recall_rate)]


def test_ap_small(kat):
    p, a = np.array([.9, .8, .7, .6]), np.array([1, 0, 1, 0])
    for f in (om.ap_at_n, apc.AveragePrecisionCalculator.ap_at_n):
        assert f(p, a, n=None) == pytest.approx(kat["C1_ap"], abs=1e-15)
        assert f(p, a, n=2) == pytest.approx(kat["C2_ap_at_2"], abs=1e-15)
        assert f(p, a, n=None, total_num_positives=4) == pytest.approx(kat["C3_ap_tot4"], abs=1e-15)
    assert apc.AveragePrecisionCalculator.ap(p, a) == pytest.approx(kat["C1_ap"], abs=1e-15)


@pytest.mark.parametrize("name,gap,hit,perr", IMPLS)
def test_c4(kat, name, gap, hit, perr):
    p = np.array(kat["C4"]["p"], dtype=np.float32)
    y = np.array(kat["C4"]["y"], dtype=np.float32)
    assert gap(p, y, 2) == pytest.approx(kat["C4"]["gap_top2"], **GAP_TOL)
    assert hit(p, y) == pytest.approx(kat["C4"]["hit1"], **GAP_TOL)
    assert perr(p, y) == pytest.approx(kat["C4"]["perr"], **GAP_TOL)


@pytest.mark.parametrize("name,gap,hit,perr", IMPLS)
def test_c5_teacher_data(kat, name, gap, hit, perr):
    rs = np.random.RandomState(1234)
    W = rs.randn(32, 4716)
    z = rs.randn(256, 32).astype(np.float32)
    logit = (z @ W / np.sqrt(32) - 3)
    yy = (logit + 0.5 * rs.randn(256, 4716) > 0.2).astype(np.float32)
    pp = (1 / (1 + np.exp(-logit))).astype(np.float32)
    assert yy.sum(1).mean() == pytest.approx(kat["C5"]["mean_labels"])
    assert gap(pp, yy, 20) == pytest.approx(kat["C5"]["gap20"], **GAP_TOL)
    assert hit(pp, yy) == pytest.approx(kat["C5"]["hit1"], **GAP_TOL)
    if name == "product":  # the loop oracle is slow for PERR at this size; product is vectorised
        assert perr(pp, yy) == pytest.approx(kat["C5"]["perr"], **GAP_TOL)


@pytest.mark.parametrize("name,gap,hit,perr", IMPLS)
def test_c6_uniform(kat, name, gap, hit, perr):
    rs = np.random.RandomState(0)
    p6 = rs.rand(64, 4716).astype(np.float32)
    y6 = (rs.rand(64, 4716) > 0.999).astype(np.float32)
    assert gap(p6, y6, 20) == pytest.approx(kat["C6"]["gap20"], **GAP_TOL)
    assert hit(p6, y6) == pytest.approx(kat["C6"]["hit1"], **GAP_TOL)
    assert perr(p6, y6) == pytest.approx(kat["C6"]["perr"], **GAP_TOL)


@pytest.mark.parametrize("name,gap,hit,perr", IMPLS)
def test_c8_small_cases(kat, name, gap, hit, perr):
    for c in kat["C8"]:
        rs = np.random.RandomState(c["seed"])
        pc = rs.rand(c["B"], c["V"]).astype(np.float32)
        yc = (rs.rand(c["B"], c["V"]) < c["dens"])
        yc[:, 0] |= (yc.sum(1) == 0)
        ycf = yc.astype(np.float32)
        assert gap(pc, ycf, c["k"]) == pytest.approx(c["gap"], **GAP_TOL), c
        assert hit(pc, ycf) == pytest.approx(c["hit1"], **GAP_TOL), c
        assert perr(pc, ycf) == pytest.approx(c["perr"], **GAP_TOL), c


def test_c7_dequantize_endpoints(kat):
    assert np_ref.dequantize(np.array([0], dtype=np.uint8))[0] == kat["C7"]["deq0"] == -1.9921875
    assert np_ref.dequantize(np.array([255], dtype=np.uint8))[0] == pytest.approx(kat["C7"]["deq255"], abs=1e-15)


def test_c9_evaluation_metrics(kat):
    rs = np.random.RandomState(kat["C9"]["seed"])
    em = eu.EvaluationMetrics(50, 20)
    for b in range(3):
        pb = rs.rand(8, 50).astype(np.float32)
        yb = (rs.rand(8, 50) < 0.1)
        yb[:, 0] |= (yb.sum(1) == 0)
        lb = float(rs.rand())
        assert lb == pytest.approx(kat["C9"]["losses"][b])
        em.accumulate(pb, yb.astype(np.float32), lb)
    res = em.get()
    assert res["avg_hit_at_one"] == pytest.approx(kat["C9"]["avg_hit_at_one"], **GAP_TOL)
    assert res["avg_perr"] == pytest.approx(kat["C9"]["avg_perr"], **GAP_TOL)
    assert res["avg_loss"] == pytest.approx(kat["C9"]["avg_loss"], **GAP_TOL)
    assert res["gap"] == pytest.approx(kat["C9"]["gap"], **GAP_TOL)
    assert float(np.mean(res["aps"])) == pytest.approx(kat["C9"]["map"], **GAP_TOL)
    em.clear()
    with pytest.raises(ValueError):
        em.get()


def test_c10_map(kat):
    rs = np.random.RandomState(kat["C10"]["seed"])
    mc = mapc.MeanAveragePrecisionCalculator(6)
    pm = rs.rand(12, 6)
    ym = (rs.rand(12, 6) < 0.4).astype(np.float64)
    mc.accumulate([pm[:, i] for i in range(6)], [ym[:, i] for i in range(6)], [None] * 6)
    assert mc.peek_map_at_n() == pytest.approx(kat["C10"]["aps"], **GAP_TOL)
    assert not mc.is_empty()
    mc.clear()
    assert mc.is_empty()


def test_metric_error_behaviour():
    with pytest.raises(ValueError):
        apc.AveragePrecisionCalculator(top_n=-1)
    with pytest.raises(ValueError):
        apc.AveragePrecisionCalculator.ap_at_n(np.zeros(3), np.zeros(4))
    with pytest.raises(ValueError):
        apc.AveragePrecisionCalculator.ap_at_n(np.zeros(3), np.zeros(3), n=0)
    with pytest.raises(ValueError):
        mapc.MeanAveragePrecisionCalculator(1)
    with pytest.raises(ValueError):
        eu.top_k_by_class(np.zeros((2, 3)), np.zeros((2, 3)), k=0)
    c = apc.AveragePrecisionCalculator()
    assert c.peek_ap_at_n() == 0
    with pytest.raises(ValueError):
        c.accumulate(np.zeros(3), np.zeros(3), num_positives=-1)
    # all-negative list -> 0
    assert apc.AveragePrecisionCalculator.ap(np.array([.3, .2]), np.array([0, 0])) == 0
