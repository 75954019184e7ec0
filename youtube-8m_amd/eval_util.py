"""Evaluation metrics (Hit@1, PERR, GAP@k); API mirrors W/eval_util.py:28-254.

Host functions take numpy arrays like the reference.  ``EvaluationMetrics.accumulate_device`` is the
MI355X path: the per-video top-k selection and the PERR rank counts run on the GPU
(yt8m_topk_rows, yt8m_perr_rows).
"""
import numpy

from . import average_precision_calculator as ap_calculator
from . import mean_average_precision_calculator as map_calculator


def flatten(l):
    return [item for sublist in l for item in sublist]


def calculate_hit_at_one(predictions, actuals):
    """W/eval_util.py:28-42."""
    top_prediction = numpy.argmax(predictions, 1)
    hits = actuals[numpy.arange(actuals.shape[0]), top_prediction]
    return numpy.average(hits)


def calculate_precision_at_equal_recall_rate(predictions, actuals):
    """W/eval_util.py:74-99, vectorised: per video, among the top-(#labels) classes, the fraction that are true
    labels with a strictly positive score."""
    num_videos = actuals.shape[0]
    order = numpy.argsort(-predictions, axis=1, kind="stable")
    num_labels = numpy.sum(actuals, axis=1).astype(numpy.int64)
    agg = 0.0
    for row in range(num_videos):
        nl = int(num_labels[row])
        if nl == 0:
            # reference: argpartition(-0)[-0:] selects ALL classes; precision over all of them
            top = numpy.arange(predictions.shape[1])
        else:
            top = _top_indices(predictions[row], nl, order[row])
        sel = predictions[row, top] > 0
        agg += float(numpy.sum(actuals[row, top][sel])) / top.size
    return agg / num_videos


def _top_indices(pred_row, n, order_row=None):
    """Same SET as numpy.argpartition(pred_row, -n)[-n:] whenever the n-th and (n+1)-th scores differ."""
    return numpy.argpartition(pred_row, -n)[-n:]


def top_k_pairs(predictions, labels, k=20):
    """Pooled per-video top-k (score, label) pairs + total positives; what calculate_gap feeds the AP
    calculator with (W/eval_util.py:102-165), without the per-class regrouping it does not need."""
    if k <= 0:
        raise ValueError("k must be a positive integer.")
    k = min(k, predictions.shape[1])
    idx = numpy.argpartition(predictions, -k, axis=1)[:, -k:]
    rows = numpy.arange(predictions.shape[0])[:, None]
    return predictions[rows, idx].reshape(-1), labels[rows, idx].reshape(-1), float(numpy.sum(labels))


def calculate_gap(predictions, actuals, top_k=20):
    """W/eval_util.py:102-120."""
    gap_calculator = ap_calculator.AveragePrecisionCalculator()
    sp, sl, npos = top_k_pairs(predictions, actuals, top_k)
    gap_calculator.accumulate(sp, sl, npos)
    return gap_calculator.peek_ap_at_n()


def top_k_by_class(predictions, labels, k=20):
    """W/eval_util.py:123-156: per-class lists of the (score, label) pairs that made some video's top-k."""
    if k <= 0:
        raise ValueError("k must be a positive integer.")
    k = min(k, predictions.shape[1])
    num_classes = predictions.shape[1]
    idx = numpy.argpartition(predictions, -k, axis=1)[:, -k:]
    out_predictions = [[] for _ in range(num_classes)]
    out_labels = [[] for _ in range(num_classes)]
    for v in range(predictions.shape[0]):
        for c in idx[v]:
            out_predictions[c].append(predictions[v, c])
            out_labels[c].append(labels[v, c])
    out_true_positives = [numpy.sum(labels[:, i]) for i in range(num_classes)]
    return out_predictions, out_labels, out_true_positives


class EvaluationMetrics(object):
    """Running Hit@1 / PERR / loss / GAP@k / per-class AP over the batches of an evaluation run; the interface
    (constructor, accumulate, get, clear and the dictionary keys) is the one W/eval_util.py:167-254 exposes to
    train.py / eval.py.

    Host and device batches end in the same place, ``_ingest``: the pooled per-video top-k (score, label, class)
    triplets feed the global AP calculator (GAP@k) and, grouped by class with one stable sort, the per-class
    calculators (mAP) -- the reference's per-class Python lists (top_k_by_class) are never built on this path.
    ``accumulate_device`` produces the triplets on the MI355X (yt8m_topk_rows, yt8m_perr_rows) so only
    B*k triplets, B PERR values and V class counts cross PCIe instead of two [B, V] matrices per step
    (W/train.py:574-575)."""

    _KEYS = ("hit_at_one", "perr", "loss")

    def __init__(self, num_class, top_k):
        self.num_class = int(num_class)
        self.top_k = top_k
        self.map_calculator = map_calculator.MeanAveragePrecisionCalculator(num_class)
        self.global_ap_calculator = ap_calculator.AveragePrecisionCalculator()
        self.clear()

    def clear(self):
        self._totals = dict.fromkeys(self._KEYS, 0.0)       # batch-size weighted sums
        self.num_examples = 0
        self.map_calculator.clear()
        self.global_ap_calculator.clear()

    # kept as attributes: W/eval.py and tests read them
    sum_hit_at_one = property(lambda self: self._totals["hit_at_one"])
    sum_perr = property(lambda self: self._totals["perr"])
    sum_loss = property(lambda self: self._totals["loss"])

    def _ingest(self, scores, hits, classes, class_positives, n, sums):
        """scores / hits / classes: flat arrays of the pooled top-k triplets; class_positives [V]: positives per class over
        ALL labels of the batch (not only the top-k ones); sums: batch totals of hit@1, PERR, loss."""
        scores = numpy.asarray(scores).reshape(-1)
        hits = numpy.asarray(hits).reshape(-1)
        classes = numpy.asarray(classes).reshape(-1)
        class_positives = numpy.asarray(class_positives).reshape(-1)
        self.global_ap_calculator.accumulate(scores, hits, float(class_positives.sum()))
        self.map_calculator.accumulate_sparse(classes, scores, hits, class_positives)
        self.num_examples += int(n)
        for key in self._KEYS:
            self._totals[key] += float(sums[key])
        return {key: float(sums[key]) / n for key in self._KEYS}

    def accumulate(self, predictions, labels, loss):
        """Host batch: predictions, labels numpy [B, V]; loss a scalar or per-example array."""
        n = labels.shape[0]
        k = min(self.top_k, predictions.shape[1])
        if k <= 0:
            raise ValueError("k must be a positive integer.")
        idx = numpy.argpartition(predictions, -k, axis=1)[:, -k:]
        rows = numpy.arange(n)[:, None]
        sums = {"hit_at_one": calculate_hit_at_one(predictions, labels) * n,
                "perr": calculate_precision_at_equal_recall_rate(predictions, labels) * n,
                "loss": float(numpy.mean(loss)) * n}
        return self._ingest(predictions[rows, idx], labels[rows, idx], idx, numpy.sum(labels, axis=0), n, sums)

    def accumulate_device(self, predictions, labels, loss, group=None):
        """Device batch: predictions, labels are tensors [B, V] on the MI355X (this rank's shard under data parallelism).
        Top-k, Hit@1, PERR and the per-class label counts are computed there."""
        import torch
        from . import ops
        vals, idx = ops.topk_rows(predictions, self.top_k)
        idx = idx.long().clamp_(0, predictions.shape[1] - 1)       # all-NaN rows report index INT_MAX: keep gather in range
        lab_u8 = labels if labels.dtype in (torch.bool, torch.uint8) else (labels > 0)
        picked = torch.gather(lab_u8.to(torch.float32), 1, idx)
        perr = ops.perr_rows(predictions, lab_u8)
        class_pos = lab_u8.to(torch.float64).sum(dim=0) if lab_u8.dtype != torch.bool else lab_u8.sum(dim=0, dtype=torch.float64)
        return self.accumulate_topk(vals, picked, idx, class_pos, perr.sum(), float(loss), group=group)

    def accumulate_topk(self, vals, picked, idx, class_positives, perr_sum, loss, group=None):
        """vals / picked / idx [B, k]: per-video top-k scores (descending), their labels and class ids; class_positives [V]:
        positives per class in the shard; perr_sum: sum of the per-video PERR values.  Under torch.distributed
        every rank's triplets are all-gathered (B*k*12 bytes per rank) and the class counts / sums all-reduced, so every
        rank accumulates the metrics of the GLOBAL batch."""
        import torch
        import torch.distributed as dist
        bs = vals.shape[0]
        dev = vals.device
        head = torch.stack([picked[:, 0].to(torch.float64).sum(), perr_sum.to(torch.float64).reshape(()),
                            torch.tensor(float(loss) * bs, dtype=torch.float64, device=dev),
                            torch.tensor(float(bs), dtype=torch.float64, device=dev)])
        stats = torch.cat([head, class_positives.to(torch.float64).reshape(-1)])
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            world = dist.get_world_size(group)
            counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(counts, torch.tensor([bs], dtype=torch.int64, device=dev), group=group)
            counts = [int(c.item()) for c in counts]
            pad = torch.zeros((max(counts), 3, vals.shape[1]), dtype=torch.float32, device=dev)
            pad[:bs, 0], pad[:bs, 1], pad[:bs, 2] = vals, picked, idx.to(torch.float32)   # class ids < 2^24: exact in fp32
            gathered = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(gathered, pad, group=group)
            vals, picked, idx = (torch.cat([g[:c, j] for g, c in zip(gathered, counts)]) for j in range(3))
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        stats = stats.cpu().numpy()
        hit_sum, perr_tot, loss_sum, n = (float(v) for v in stats[:4])
        return self._ingest(vals.cpu().numpy(), picked.cpu().numpy(), idx.cpu().numpy().astype(numpy.int64), stats[4:], int(n),
                            {"hit_at_one": hit_sum, "perr": perr_tot, "loss": loss_sum})

    def get(self):
        if self.num_examples <= 0:
            raise ValueError("total_sample must be positive.")
        out = {"avg_" + key: self._totals[key] / self.num_examples for key in self._KEYS}
        out["aps"] = self.map_calculator.peek_map_at_n()
        out["gap"] = self.global_ap_calculator.peek_ap_at_n()
        return out
