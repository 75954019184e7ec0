"""Per-class average precision (interface of W/mean_average_precision_calculator.py:44-112)."""
from . import average_precision_calculator


class MeanAveragePrecisionCalculator(object):
    def __init__(self, num_class):
        if not isinstance(num_class, int) or num_class <= 1:
            raise ValueError("num_class must be a positive integer.")
        self._num_class = num_class
        self._ap_calculators = [average_precision_calculator.AveragePrecisionCalculator() for _ in range(num_class)]

    def accumulate(self, predictions, actuals, num_positives=None):
        if not num_positives:
            num_positives = [None for _ in range(len(predictions))]
        for i in range(len(predictions)):
            self._ap_calculators[i].accumulate(predictions[i], actuals[i], num_positives[i])

    def accumulate_sparse(self, class_ids, predictions, actuals, num_positives):
        """Flat (class, score, label) triplets instead of one list per class: grouped with a single stable sort, so only the
        classes that occur are touched; num_positives [num_class] still counts every class (a class whose positives never
        reach a top-k keeps its denominator)."""
        import numpy as np
        class_ids = np.asarray(class_ids).reshape(-1)
        predictions = np.asarray(predictions).reshape(-1)
        actuals = np.asarray(actuals).reshape(-1)
        order = np.argsort(class_ids, kind="stable")
        sorted_ids = class_ids[order]
        bounds = np.flatnonzero(np.diff(sorted_ids)) + 1
        starts = np.concatenate(([0], bounds)) if len(sorted_ids) else np.zeros(0, dtype=np.int64)
        ends = np.concatenate((bounds, [len(sorted_ids)])) if len(sorted_ids) else starts
        seen = set()
        for s, e in zip(starts, ends):
            c = int(sorted_ids[s])
            seen.add(c)
            sel = order[s:e]
            self._ap_calculators[c].accumulate(predictions[sel], actuals[sel], float(num_positives[c]))
        for c in np.flatnonzero(np.asarray(num_positives) > 0):
            if int(c) not in seen:
                self._ap_calculators[int(c)].accumulate(np.zeros(0), np.zeros(0), float(num_positives[c]))

    def clear(self):
        for c in self._ap_calculators:
            c.clear()

    def is_empty(self):
        return all(c.heap_size == 0 for c in self._ap_calculators)

    def peek_map_at_n(self):
        return [c.peek_ap_at_n() for c in self._ap_calculators]
