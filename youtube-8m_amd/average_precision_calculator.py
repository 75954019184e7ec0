"""Average precision over a pooled list of (score, label) pairs -- host side of the GAP metric.

Interface mirrors W/average_precision_calculator.py:61-253 (accumulate / peek_ap_at_n / ap / ap_at_n / clear)
but keeps flat numpy arrays instead of a Python heap and ranks with one vectorised stable sort.
Tie order: the reference shuffles with random.seed(0); random.sample(range(n), n) before a stable sort
(:248-253); that permutation is reproduced so results match on tied scores too (for the running
interpreter's `random`; see SURVEY.md Appendix C caveat).
"""
import numbers
import random

import numpy as np


class AveragePrecisionCalculator(object):
    def __init__(self, top_n=None):
        if not ((isinstance(top_n, int) and top_n >= 0) or top_n is None):
            raise ValueError("top_n must be a positive integer or None.")
        self._top_n = top_n
        self._total_positives = 0
        self._pred = []   # list of 1-D arrays
        self._act = []

    @property
    def heap_size(self):
        n = int(sum(len(a) for a in self._pred))
        return n if self._top_n is None else min(n, self._top_n)

    @property
    def num_accumulated_positives(self):
        return self._total_positives

    def accumulate(self, predictions, actuals, num_positives=None):
        predictions = np.asarray(predictions).reshape(-1)
        actuals = np.asarray(actuals).reshape(-1)
        if len(predictions) != len(actuals):
            raise ValueError("the shape of predictions and actuals does not match.")
        if num_positives is not None:
            if not isinstance(num_positives, numbers.Number) or num_positives < 0:
                raise ValueError("'num_positives' was provided but it wan't a nonzero number.")
            self._total_positives += num_positives
        else:
            self._total_positives += int(np.count_nonzero(actuals > 0))
        self._pred.append(predictions)
        self._act.append(actuals)
        if self._top_n is not None:
            self._compact()

    def _compact(self):
        p = np.concatenate(self._pred) if self._pred else np.zeros(0)
        a = np.concatenate(self._act) if self._act else np.zeros(0)
        if len(p) > self._top_n:
            keep = np.argpartition(-p, self._top_n - 1)[:self._top_n]
            p, a = p[keep], a[keep]
        self._pred, self._act = [p], [a]

    def clear(self):
        self._pred, self._act = [], []
        self._total_positives = 0

    def peek_ap_at_n(self):
        if self.heap_size <= 0:
            return 0
        p = np.concatenate(self._pred)
        a = np.concatenate(self._act)
        return self.ap_at_n(p, a, n=self._top_n, total_num_positives=self._total_positives)

    @staticmethod
    def ap(predictions, actuals):
        return AveragePrecisionCalculator.ap_at_n(predictions, actuals, n=None)

    @staticmethod
    def ap_at_n(predictions, actuals, n=20, total_num_positives=None):
        if len(predictions) != len(actuals):
            raise ValueError("the shape of predictions and actuals does not match.")
        if n is not None:
            if not isinstance(n, int) or n <= 0:
                raise ValueError("n must be 'None' or a positive integer. It was '%s'." % n)
        predictions = np.asarray(predictions)
        actuals = np.asarray(actuals)
        m = len(predictions)
        random.seed(0)
        perm = np.asarray(random.sample(range(m), m), dtype=np.int64)
        predictions, actuals = predictions[perm], actuals[perm]
        order = np.argsort(-predictions, kind="stable")       # descending, ties in shuffled order
        numpos = int(np.count_nonzero(actuals > 0)) if total_num_positives is None else total_num_positives
        if numpos == 0:
            return 0
        if n is not None:
            numpos = min(numpos, n)
        r = m if n is None else min(m, n)
        rel = (actuals[order[:r]] > 0)
        if not rel.any():
            return 0.0
        hits = np.cumsum(rel)
        ranks = np.arange(1, r + 1)
        return float(np.sum((hits[rel] / ranks[rel])) / numpos)
