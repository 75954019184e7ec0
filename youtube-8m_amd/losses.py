"""Label losses; names, flags and semantics mirror W/losses.py (W = /root/reference/youtube-8m-wangheda)."""
import torch

from . import ops
from .flags import FLAGS, DEFINE_float, DEFINE_string, DEFINE_integer, DEFINE_bool

# W/losses.py:22-44
DEFINE_integer("num_classes", 4716, "number of classes")
DEFINE_float("support_loss_percent", 0.1, "the part that support loss (in multi-task scenario) take in the whole loss function.")
DEFINE_string("support_type", "vertical", "type of support label, vertical or frequent or vertical,frequent.")
DEFINE_integer("num_supports", 25, "Number of total support categories.")
DEFINE_integer("num_verticals", 25, "Number of total vertical categories.")
DEFINE_integer("num_frequents", 200, "Number of total frequent categories.")
DEFINE_string("vertical_file", "resources/vertical.tsv", "Location of label-vertical mapping file.")
DEFINE_bool("label_smoothing", False, "whether do label smoothing")
DEFINE_float("label_smoothing_epsilon", 0.1, "whether do label smoothing")


def smoothing(labels):
    """W/losses.py:46-54: y*(1-eps) + (sum_l y / K)*eps.  (Label preparation: not differentiated.)"""
    epsilon = FLAGS.label_smoothing_epsilon
    y = labels.to(torch.float32)
    prior = y.sum(dim=1, keepdim=True) / y.shape[1]
    return y * (1.0 - epsilon) + prior * epsilon


def load_vertical_mapping(path, num_classes, num_verticals):
    """W/losses.py:233-243: every line holding exactly two integers "class vertical" sets vm[class, vertical] = 1; other lines
    are skipped (a non-integer token raises, as in the reference).  Returns a float32 numpy array [num_classes, num_verticals]."""
    import numpy as np
    vm = np.zeros((num_classes, num_verticals), dtype=np.float32)
    with open(path) as fh:
        for line in fh:
            group = [int(t) for t in line.strip().split()]
            if len(group) == 2:
                x, y = group
                vm[x, y] = 1
    return vm


_VERTICAL_MAPPINGS = {}      # (file, num_classes, num_verticals, device) -> device tensor (the reference's untrainable variable "vm")


def _vertical_mapping(device):
    key = (FLAGS.vertical_file, FLAGS.num_classes, FLAGS.num_verticals, str(device))
    vm = _VERTICAL_MAPPINGS.get(key)
    if vm is None:
        vm = torch.from_numpy(load_vertical_mapping(FLAGS.vertical_file, FLAGS.num_classes, FLAGS.num_verticals)).to(device)
        _VERTICAL_MAPPINGS[key] = vm
    return vm


class BaseLoss(object):
    """W/losses.py:56-73."""

    def calculate_loss(self, unused_predictions, unused_labels, **unused_params):
        raise NotImplementedError()


class CrossEntropyLoss(BaseLoss):
    """W/losses.py:110-130: probability-space cross entropy, epsilon = 10e-6, sum over classes, mean over batch,
    optional per-example weights."""

    def calculate_loss(self, predictions, labels, weights=None, scale=1.0, **unused_params):
        y = smoothing(labels) if FLAGS.label_smoothing else labels
        return ops.cross_entropy(predictions, y, weights, scale)


class MultiTaskLoss(BaseLoss):
    """W/losses.py:216-257 (virtual)."""

    def calculate_loss(self, unused_predictions, unused_labels, **unused_params):
        raise NotImplementedError()

    def get_support(self, labels, support_type=None):
        if support_type is None:
            support_type = FLAGS.support_type
        if "," in support_type:
            return torch.cat([self.get_support(labels, st).to(torch.float32) for st in support_type.split(",")], dim=1)
        if support_type == "label":
            return labels.to(torch.float32)
        if support_type == "frequent":
            return labels[:, :FLAGS.num_frequents].to(torch.float32)
        if support_type == "vertical":
            # W/losses.py:229-246: labels . vm > 0.2 with vm the 0/1 class -> vertical table of --vertical_file (the file itself
            # comes from the reference's eda/ tooling, SURVEY.md 2.1: supply it; a missing file raises here as open() does there)
            float_labels = labels.to(torch.float32).contiguous()
            if float_labels.shape[1] != FLAGS.num_classes:
                raise ValueError("labels have %d classes, --num_classes is %d" % (float_labels.shape[1], FLAGS.num_classes))
            vertical_labels = ops.gemm(float_labels, _vertical_mapping(labels.device))
            return (vertical_labels > 0.2).to(torch.float32)
        raise NotImplementedError()


class MultiTaskCrossEntropyLoss(MultiTaskLoss):
    """W/losses.py:271-279: (1-s)*CE(pred, y) + s*CE(support_pred, support(y))."""

    def calculate_loss(self, predictions, support_predictions, labels, **unused_params):
        support_labels = self.get_support(labels)
        s = FLAGS.support_loss_percent
        ce = CrossEntropyLoss()
        kw = {k: v for k, v in unused_params.items() if k != "scale"}
        return (ce.calculate_loss(predictions, labels, scale=1.0 - s, **kw)
                + ce.calculate_loss(support_predictions, support_labels, scale=s, **kw))
