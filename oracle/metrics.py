"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- restatement of the reference's host metrics.

PINNED: checked against known answers produced by the imported reference code
(tests/golden/make_golden.py -> tests/golden/metrics_kat.json), see tests/test_oracle_metrics.py.

Follows W/eval_util.py:28-165 and W/average_precision_calculator.py:93-253 (paths relative to
/root/reference/youtube-8m-wangheda/).  Written loop-for-loop (slow, obviously-correct) on purpose;
the product's vectorised implementation lives in youtube-8m_amd/eval_util.py.

Tie caveat: the reference shuffles with ``random.seed(0);
random.sample(range(n), n)`` before a stable sort (average_precision_calculator.py:248-253), so the
order of exactly-tied scores depends on the Python version's `random`.  The oracle reproduces the
shuffle with the *running* interpreter's `random`, which is what the golden vectors were made with.
"""
import random

import numpy as np


def hit_at_one(predictions, actuals):
    """W/eval_util.py:28-42."""
    total = 0.0
    for r in range(predictions.shape[0]):
        total += float(actuals[r, int(np.argmax(predictions[r]))])
    return total / predictions.shape[0]


def precision_at_equal_recall_rate(predictions, actuals):
    """W/eval_util.py:74-99: for each video take the top-(#labels) classes; precision among those with
    a strictly positive score; mean over videos."""
    agg = 0.0
    nv = actuals.shape[0]
    for r in range(nv):
        nl = int(np.sum(actuals[r]))
        top = np.argpartition(predictions[r], -nl)[-nl:]
        item = 0.0
        for li in top:
            if predictions[r][li] > 0:
                item += actuals[r][li]
        item /= top.size
        agg += item
    return agg / nv


def _shuffle(pred, act):
    """average_precision_calculator.py:248-253."""
    random.seed(0)
    idx = random.sample(range(len(pred)), len(pred))
    return pred[idx], act[idx]


def ap_at_n(predictions, actuals, n=20, total_num_positives=None):
    """average_precision_calculator.py:172-245."""
    predictions = np.array(predictions)
    actuals = np.array(actuals)
    if len(predictions) != len(actuals):
        raise ValueError("the shape of predictions and actuals does not match.")
    if n is not None and (not isinstance(n, int) or n <= 0):
        raise ValueError("n must be 'None' or a positive integer. It was '%s'." % n)
    predictions, actuals = _shuffle(predictions, actuals)
    order = sorted(range(len(predictions)), key=lambda k: predictions[k], reverse=True)
    numpos = np.size(np.where(actuals > 0)) if total_num_positives is None else total_num_positives
    if numpos == 0:
        return 0
    if n is not None:
        numpos = min(numpos, n)
    delta_recall = 1.0 / numpos
    poscount = 0.0
    ap = 0.0
    r = len(order) if n is None else min(len(order), n)
    for i in range(r):
        if actuals[order[i]] > 0:
            poscount += 1
            ap += poscount / (i + 1) * delta_recall
    return ap


def top_k_pairs(predictions, actuals, k=20):
    """W/eval_util.py:123-165 reduced to what calculate_gap consumes: the per-video top-k
    (score, label) pairs pooled over videos, and the total number of positives (ALL labels, not just
    the ones inside the top-k)."""
    k = min(k, predictions.shape[1])
    sp, sl = [], []
    for r in range(predictions.shape[0]):
        idx = np.argpartition(predictions[r], -k)[-k:]
        for i in idx:
            sp.append(predictions[r][i])
            sl.append(actuals[r][i])
    return np.array(sp), np.array(sl), float(np.sum(actuals))


def gap(predictions, actuals, top_k=20):
    """W/eval_util.py:102-120: global average precision over pooled per-video top-k pairs."""
    sp, sl, npos = top_k_pairs(predictions, actuals, top_k)
    if len(sp) == 0:
        return 0
    return ap_at_n(sp, sl, n=None, total_num_positives=npos)
