"""Evaluation metrics (Hit@1, PERR, GAP@k); API mirrors W/eval_util.py:28-254.

Host functions take numpy arrays like the reference.  ``EvaluationMetrics.accumulate_device`` is the
MI355X path: the per-video top-k selection runs on the GPU (yt8m_topk_rows) and only
B*k (score, label) pairs plus B hit / PERR scalars cross PCIe, instead of two [B, 4716] matrices per step
(W/train.py:574-575).
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
    """W/eval_util.py:167-254."""

    def __init__(self, num_class, top_k):
        self.sum_hit_at_one = 0.0
        self.sum_perr = 0.0
        self.sum_loss = 0.0
        self.map_calculator = map_calculator.MeanAveragePrecisionCalculator(num_class)
        self.global_ap_calculator = ap_calculator.AveragePrecisionCalculator()
        self.top_k = top_k
        self.num_examples = 0

    def accumulate(self, predictions, labels, loss):
        batch_size = labels.shape[0]
        mean_hit_at_one = calculate_hit_at_one(predictions, labels)
        mean_perr = calculate_precision_at_equal_recall_rate(predictions, labels)
        mean_loss = numpy.mean(loss)
        sparse_predictions, sparse_labels, num_positives = top_k_by_class(predictions, labels, self.top_k)
        self.map_calculator.accumulate(sparse_predictions, sparse_labels, num_positives)
        self.global_ap_calculator.accumulate(flatten(sparse_predictions), flatten(sparse_labels), sum(num_positives))
        self.num_examples += batch_size
        self.sum_hit_at_one += mean_hit_at_one * batch_size
        self.sum_perr += mean_perr * batch_size
        self.sum_loss += mean_loss * batch_size
        return {"hit_at_one": mean_hit_at_one, "perr": mean_perr, "loss": mean_loss}

    def accumulate_device(self, predictions, labels, loss, group=None):
        """GPU path for GAP / Hit@1: predictions, labels are device tensors [B, V] (this rank's shard under data parallelism).
        Per-video top-k runs on the device (yt8m_topk_rows); only B*k (score, label) pairs leave it."""
        import torch
        from . import ops
        vals, idx = ops.topk_rows(predictions, self.top_k)
        lab = labels.to(torch.float32)
        picked = torch.gather(lab, 1, idx.long())
        return self.accumulate_topk(vals, picked, lab.sum(), float(loss), group=group)

    def accumulate_topk(self, vals, picked, num_positives, loss, group=None):
        """vals / picked [B, k]: per-video top-k scores (descending) and their labels; num_positives: scalar tensor = number of
        positive labels in the shard (ALL labels, not only the top-k ones: W/average_precision_calculator.py total_positives).
        Under torch.distributed every rank's pairs are all-gathered (B*k*8 bytes per rank) and the positives /
        loss sums all-reduced, so every rank accumulates the metrics of the GLOBAL batch."""
        import torch
        import torch.distributed as dist
        bs = vals.shape[0]
        stats = torch.stack([num_positives.to(torch.float64).reshape(()), picked[:, 0].to(torch.float64).sum(),
                             torch.tensor(float(loss) * bs, dtype=torch.float64, device=vals.device),
                             torch.tensor(float(bs), dtype=torch.float64, device=vals.device)])
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            world = dist.get_world_size(group)
            counts = [torch.zeros(1, dtype=torch.int64, device=vals.device) for _ in range(world)]
            dist.all_gather(counts, torch.tensor([bs], dtype=torch.int64, device=vals.device), group=group)
            cap = int(max(int(c.item()) for c in counts))
            pad = torch.zeros((cap, 2, vals.shape[1]), dtype=torch.float32, device=vals.device)
            pad[:bs, 0], pad[:bs, 1] = vals, picked
            gathered = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(gathered, pad, group=group)
            vals = torch.cat([g[:int(c.item()), 0] for g, c in zip(gathered, counts)])
            picked = torch.cat([g[:int(c.item()), 1] for g, c in zip(gathered, counts)])
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        npos, hits, loss_sum, n = (float(v) for v in stats.cpu())
        self.global_ap_calculator.accumulate(vals.reshape(-1).cpu().numpy(), picked.reshape(-1).cpu().numpy(), npos)
        self.num_examples += int(n)
        self.sum_hit_at_one += hits
        self.sum_loss += loss_sum
        return {"hit_at_one": hits / n, "loss": loss_sum / n}

    def get(self):
        if self.num_examples <= 0:
            raise ValueError("total_sample must be positive.")
        avg_hit_at_one = self.sum_hit_at_one / self.num_examples
        avg_perr = self.sum_perr / self.num_examples
        avg_loss = self.sum_loss / self.num_examples
        aps = self.map_calculator.peek_map_at_n()
        gap = self.global_ap_calculator.peek_ap_at_n()
        return {"avg_hit_at_one": avg_hit_at_one, "avg_perr": avg_perr, "avg_loss": avg_loss, "aps": aps, "gap": gap}

    def clear(self):
        self.sum_hit_at_one = 0.0
        self.sum_perr = 0.0
        self.sum_loss = 0.0
        self.map_calculator.clear()
        self.global_ap_calculator.clear()
        self.num_examples = 0
