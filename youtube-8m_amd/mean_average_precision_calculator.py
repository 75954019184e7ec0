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

    def clear(self):
        for c in self._ap_calculators:
            c.clear()

    def is_empty(self):
        return all(c.heap_size == 0 for c in self._ap_calculators)

    def peek_map_at_n(self):
        return [c.peek_ap_at_n() for c in self._ap_calculators]
