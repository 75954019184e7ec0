"""Training-step slice of W/train.py (W = /root/reference/youtube-8m-wangheda): plugin lookup
(find_class_by_name, :212-215), and build_graph (:262-479) re-expressed as an eager step:

    transform (:343-344) -> model.create_model (:361-377) -> label loss (:384-430) -> backward
    -> [RCCL mean all-reduce] -> + l2*w (:435-459) -> per-tensor clip (:463-465) -> Adam (:466)

TF queues / Supervisor / Saver / summaries (the control plane) are out of scope.
"""
import math

import torch

from . import feature_transform, losses, ops
from .feature_transform import DefaultTransformer
from .flags import FLAGS, DEFINE_integer, DEFINE_float, DEFINE_string, DEFINE_bool
from .variables import Graph, get_default_graph, set_default_graph

# W/train.py:73-137 (only the flags the hot path reads)
DEFINE_string("model", "LogisticModel", "Which architecture to use for the model.")
DEFINE_integer("batch_size", 1024, "How many examples to process per batch for training.")
DEFINE_string("label_loss", "CrossEntropyLoss", "Which loss function to use for training the model.")
DEFINE_float("regularization_penalty", 1, "How much weight to give to the regularization loss (the label loss has a weight of 1).")
DEFINE_float("base_learning_rate", 0.01, "Which learning rate to start with.")
DEFINE_float("learning_rate_decay", 0.95, "Learning rate decay factor to be applied every learning_rate_decay_examples.")
DEFINE_float("learning_rate_decay_examples", 4000000, "Multiply current learning rate by learning_rate_decay every learning_rate_decay_examples.")
DEFINE_string("optimizer", "AdamOptimizer", "What optimizer class to use.")
DEFINE_float("clip_gradient_norm", 1.0, "Norm to clip gradients to.")
DEFINE_bool("multitask", False, "Whether to consider support_predictions")
DEFINE_string("feature_names", "mean_rgb", "Name of the feature to use for training.")
DEFINE_string("feature_sizes", "1024", "Length of the feature vectors.")
# new: raw uint8 frame blocks reach models that fold the input transform into their first GEMM (NetVLAD)
DEFINE_bool("fold_dequant", True, "Hand raw uint8 frames to models that declare accepts_quantized_input.")
# W/train.py:53-64 distillation inputs: a second model's predictions arrive with the batch
DEFINE_bool("distillation_features", False, "If set, *DistillationFeatureReader will be used, the feature must contains the "
            "added distillation_predictions features.")
DEFINE_integer("distillation_type", 0, "Type of distillation, options are 1 and 2.")
DEFINE_bool("distillation_as_input", False, "If set true, distillation_predictions will be given to model.")
DEFINE_bool("distillation_as_boosting", False, "If set true, distillation_predictions will be used in computation of weighted loss.")
DEFINE_float("distillation_percent", 0.0, "If larger than 0, final_loss = distillation_loss * percent + normal_loss * (1.0 - percent).")
DEFINE_bool("dropout", False, "Whether to consider dropout")
DEFINE_float("keep_prob", 1.0, "probability to keep output (used in dropout, keep it unchanged in validationg and test)")
DEFINE_float("noise_level", 0.0, "standard deviation of noise (added to hidden nodes)")
DEFINE_bool("frame_features", False, "If set, then --train_data_pattern must be frame-level features.")


def find_class_by_name(name, modules):
    """W/train.py:212-215: the first module that has the attribute wins; a missing class raises StopIteration."""
    modules = [getattr(module, name, None) for module in modules]
    return next(a for a in modules if a)


def exponential_decay(base_lr, global_step, batch_size, decay_examples, decay):
    """tf.train.exponential_decay(staircase=True) as called at W/train.py:303-308."""
    return base_lr * decay ** math.floor(global_step * batch_size / float(decay_examples))


def reform_distill_labels(labels_batch, distill_labels_batch, p):
    """W/train.py:320-327 (distillation_type == 2): labels + distill * (sum(labels) / (sum(distill) + 1e-6) * p), clipped to
    [0, 1].  Label preparation (not differentiated)."""
    float_labels = labels_batch.to(torch.float32)
    sum_float_labels = float_labels.sum(dim=1, keepdim=True)
    sum_distill_labels = distill_labels_batch.sum(dim=1, keepdim=True) + 1e-6
    return (float_labels + distill_labels_batch * (sum_float_labels / sum_distill_labels * p)).clamp_(0.0, 1.0)


def get_weights_by_predictions(labels_batch, predictions):
    """W/train.py:250-260 (distillation_as_boosting): per-video weight 3.0 where the teacher's cross entropy is above the
    batch mean, else 0.5 (epsilon 1e-6 here, unlike the loss's 1e-5)."""
    epsilon = 1e-6
    float_labels = labels_batch.to(torch.float32)
    ce = -(float_labels * torch.log(predictions + epsilon) + (1 - float_labels) * torch.log(1 - predictions + epsilon)).sum(dim=1)
    mean_ce = (ce + epsilon).mean()
    return torch.where(ce > mean_ce, torch.full_like(ce, 3.0), torch.full_like(ce, 0.5))


class TrainGraph(object):
    """What build_graph() wires, as an object with an eager ``step``."""

    def __init__(self, model, label_loss_fn=None, batch_size=1024, base_learning_rate=0.01,
                 learning_rate_decay_examples=4000000, learning_rate_decay=0.95, transformer_class=None,
                 clip_gradient_norm=1.0, regularization_penalty=1, multitask=None, graph=None, reducer=None,
                 beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.model = model
        self.label_loss_fn = label_loss_fn or losses.CrossEntropyLoss()
        self.batch_size = batch_size                      # GLOBAL batch (drives the LR staircase, :305)
        self.base_learning_rate = base_learning_rate
        self.decay_examples = learning_rate_decay_examples
        self.decay = learning_rate_decay
        self.transformer = (transformer_class or feature_transform.DefaultTransformer)()
        self.clip = clip_gradient_norm
        self.reg_penalty = regularization_penalty
        self.multitask = FLAGS.multitask if multitask is None else multitask
        self.graph = graph or get_default_graph()
        self.reducer = reducer                            # parallel.GradReducer or None
        self.global_step = 0
        self.b1, self.b2, self.eps = beta1, beta2, epsilon

    def _transform(self, model_input_raw, num_frames):
        """W/train.py:352-353 feature transform.  Raw uint8 frame blocks go to the model untouched when the model folds
        the DefaultTransformer (dequantise + zero padding + l2-normalise) into its first GEMM."""
        if (model_input_raw.dtype == torch.uint8 and model_input_raw.dim() == 3 and FLAGS.fold_dequant
                and isinstance(self.transformer, DefaultTransformer)
                and getattr(self.model, "accepts_quantized_input", False)):
            return model_input_raw, num_frames
        return self.transformer.transform(model_input_raw, num_frames=num_frames)

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, model_input_raw, labels_batch=None, num_frames=None, is_training=True, fuse_loss=True,
                distillation_predictions=None):
        g = set_default_graph(self.graph)
        g.begin_step()
        model_input, num_frames = self._transform(model_input_raw, num_frames)
        kw = {} if is_training else {"is_training": False}
        if not fuse_loss:
            kw["fuse_loss"] = False
        if FLAGS.dropout:                                         # W/train.py:359-369; the keep_prob placeholder defaults to
            kw["dropout"] = True                                  # 1.0 and is fed FLAGS.keep_prob by the training loop (:569)
            kw["keep_prob"] = float(FLAGS.keep_prob) if is_training else 1.0
        noise_level = float(FLAGS.noise_level) if (FLAGS.noise_level > 0 and is_training) else None      # :349-352, :571
        result = self.model.create_model(model_input, num_frames=num_frames, vocab_size=FLAGS.num_classes
                                         if labels_batch is None else labels_batch.shape[1],
                                         labels=labels_batch, distillation_predictions=distillation_predictions,
                                         noise_level=noise_level, **kw)
        return result

    def _label_loss(self, result, labels, weights):
        if self.multitask:                                        # W/train.py:394-413
            return self.label_loss_fn.calculate_loss(result["predictions"], result["support_predictions"], labels, weights=weights)
        return self.label_loss_fn.calculate_loss(result["predictions"], labels, weights=weights)

    def loss(self, result, labels_batch, weights=None, distill_labels_batch=None):
        """W/train.py:384-430: a model-provided "loss" wins; otherwise the label loss, optionally blended with / replaced by
        the loss against the distillation labels and weighted by the boosting weights."""
        if "loss" in result:                                      # W/train.py:384-385
            return result["loss"]
        if FLAGS.distillation_as_boosting and distill_labels_batch is not None:            # :391-392
            weights = get_weights_by_predictions(labels_batch, distill_labels_batch)
        if FLAGS.distillation_features and distill_labels_batch is not None:
            if FLAGS.distillation_type == 1:                      # :398-407 / :415-424
                p = FLAGS.distillation_percent
                if p <= 0:
                    return self._label_loss(result, labels_batch, weights)
                if p >= 1:
                    return self._label_loss(result, distill_labels_batch, weights)
                return self._label_loss(result, labels_batch, weights) * (1.0 - p) + \
                    self._label_loss(result, distill_labels_batch, weights) * p
            if FLAGS.distillation_type == 2:                      # :408-410 / :425-427 "pure distillation loss"
                return self._label_loss(result, distill_labels_batch, weights)
        return self._label_loss(result, labels_batch, weights)

    def ensure_finalized(self):
        """Freezes the variable set into the flat parameter / gradient / Adam arenas (Graph.finalize), applies
        --regularization_penalty to the per-variable l2 table and attaches the data-parallel reducer -- once.  Called by
        step() after the first forward pass and by checkpoint.restore() (the Adam slots live in the arenas)."""
        g = self.graph
        if not g.finalized:
            g.finalize()
        if not getattr(self, "_finalized_here", False):          # the l2 table is scaled exactly once per TrainGraph
            if self.reg_penalty != 1:
                g.l2.mul_(float(self.reg_penalty))
            self._finalized_here = True
        if self.reducer is not None and not getattr(self, "_reducer_attached", False):
            first = not getattr(self, "_reducer_ever_attached", False)
            self.reducer.attach(g, broadcast=first)               # its own flag: close() detaches, a later step() re-attaches
            self._reducer_attached = self._reducer_ever_attached = True     # (without a second broadcast: ADVICE r5)
            return first and getattr(self.reducer, "active", False)
        return False

    # ---- one optimisation step -----------------------------------------------------------------------
    def step(self, model_input_raw, labels_batch, num_frames=None, weights=None, distill_labels_batch=None):
        g = self.graph
        distill = distill_labels_batch if FLAGS.distillation_features else None
        if distill is not None and FLAGS.distillation_type == 2:  # W/train.py:320-327: labels are re-formed up front
            distill = reform_distill_labels(labels_batch, distill, FLAGS.distillation_percent)
        # (the fused mixing + loss of MoeModel is this build's addition: it must not pre-empt weights / distillation)
        # and it computes exactly CrossEntropyLoss: any other configured loss (TrainGraph(label_loss_fn=...), multitask) wins
        fuse = (weights is None and distill is None and not self.multitask
                and type(self.label_loss_fn) is losses.CrossEntropyLoss)
        result = self.forward(model_input_raw, labels_batch, num_frames, fuse_loss=fuse,
                              distillation_predictions=distill if FLAGS.distillation_as_input else None)
        label_loss = self.loss(result, labels_batch, weights, distill)
        # W/train.py:435-456: a model may hand back its own "regularization_loss" (added to the final loss with the
        # --regularization_penalty weight; the slim l2 regularisers are applied as l2*w inside the optimiser pass) and
        # "update_ops" (run before the gradient step, e.g. moving averages)
        final_loss = label_loss
        if "regularization_loss" in result and torch.is_tensor(result["regularization_loss"]) \
                and result["regularization_loss"].requires_grad and self.reg_penalty != 0:
            final_loss = label_loss + float(self.reg_penalty) * result["regularization_loss"]
        for op in result.get("update_ops", ()) or ():
            if callable(op):
                op()
        if self.ensure_finalized():
            # the first step of a data-parallel run: the variables were created (and the forward pass ran) BEFORE rank 0's parameters
            # arrived.  Run the forward pass again so that the whole step -- activations, weight images, backward -- sees one set of
            # weights on every rank (once per run; with identical seeds the values do not change).
            g._rng_step -= 1                                      # the same forward pass again: the same random-op seeds
            result = self.forward(model_input_raw, labels_batch, num_frames, fuse_loss=fuse,
                                  distillation_predictions=distill if FLAGS.distillation_as_input else None)
            final_loss = label_loss = self.loss(result, labels_batch, weights, distill)
            if "regularization_loss" in result and torch.is_tensor(result["regularization_loss"]) \
                    and result["regularization_loss"].requires_grad and self.reg_penalty != 0:
                final_loss = label_loss + float(self.reg_penalty) * result["regularization_loss"]
        lr = exponential_decay(self.base_learning_rate, self.global_step, self.batch_size, self.decay_examples, self.decay)
        t = self.global_step + 1
        lr_t = lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        if self.reducer is not None:
            self.reducer.begin_step()
            g.early_optimizer = None
        else:
            # single device: a recurrent stack's backward pass may apply clip + Adam to the variables whose gradients are final when
            # it starts (seq_ops._early_optimizer_hook); what it covered comes back in g.early_done
            g.early_optimizer = {"lr_t": lr_t, "clip": self.clip, "beta1": self.b1, "beta2": self.b2, "eps": self.eps}
        g.early_done = []
        try:
            final_loss.backward()
        finally:                                                # (a failed backward must not leave the pass armed)
            g.early_optimizer = None
            g.early_active = None
            g.defer_head_dw = False
        ops.join_side_work(g)                                   # gradients left on a side stream: before any optimiser pass reads them
        for v in g.trainable_variables():                       # variables the step did not touch: TF skips them
            if not v.grad_written:                              # (None gradient); here their gradient is zero
                v.grad.zero_()
                v.grad_written = True
        if self.reducer is None:
            nt, pos = len(g.trainable_variables()), 0
            for lo, hi in sorted(g.early_done) + [(nt, nt)]:    # everything the early pass did not cover
                ops.sqnorm_and_adam(g, lr_t, gscale=1.0, clip=self.clip, beta1=self.b1, beta2=self.b2, eps=self.eps, tensors=(pos, lo))
                pos = hi
            g.early_done = []
        else:
            # gradients are SUMMED over ranks bucket by bucket; the 1/world mean is folded into the optimiser pass, and
            # each bucket is clipped + updated as soon as ITS all-reduce has landed (later buckets still on the wire)
            for lo, hi in self.reducer.finished_buckets():
                ops.sqnorm_and_adam(g, lr_t, gscale=self.reducer.gscale, clip=self.clip, beta1=self.b1, beta2=self.b2,
                                    eps=self.eps, tensors=(lo, hi))
        self.global_step += 1
        return {"loss": label_loss.detach(), "predictions": result["predictions"].detach(),
                "global_step": self.global_step, "learning_rate": lr}

    def close(self):
        """End of training (where W/train.py:612-622 leaves the Supervisor's session): gives back what the step holds
        process-wide -- the reducer's CU reserve of the persistent recurrences (parallel.GradReducer.detach)."""
        if self.reducer is not None and getattr(self, "_reducer_attached", False):
            self.reducer.detach()
            self._reducer_attached = False                        # a later step() re-attaches (the l2 table is NOT rescaled)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @torch.no_grad()
    def predict(self, model_input_raw, num_frames=None, vocab_size=None):
        g = set_default_graph(self.graph)
        g.begin_step()
        model_input, num_frames = self._transform(model_input_raw, num_frames)
        result = self.model.create_model(model_input, num_frames=num_frames, vocab_size=vocab_size or FLAGS.num_classes,
                                         is_training=False)
        return result["predictions"]

    def regularization_loss(self):
        """sum_W l2 * 0.5 * |W|^2 (W/train.py:435-445); reporting only -- its gradient l2*w is applied in the
        fused optimiser pass."""
        tot = 0.0
        for v in self.graph.trainable_variables():
            if v.l2 > 0:
                tot = tot + self.reg_penalty * v.l2 * 0.5 * float((v.data.double() ** 2).sum())
        return tot


def build_graph(model, label_loss_fn=None, batch_size=None, **kw):
    """Name-compatible entry: returns the TrainGraph configured from FLAGS like W/train.py:679-728 does."""
    return TrainGraph(model, label_loss_fn=label_loss_fn,
                      batch_size=batch_size or FLAGS.batch_size,
                      base_learning_rate=kw.pop("base_learning_rate", FLAGS.base_learning_rate),
                      learning_rate_decay_examples=kw.pop("learning_rate_decay_examples", FLAGS.learning_rate_decay_examples),
                      learning_rate_decay=kw.pop("learning_rate_decay", FLAGS.learning_rate_decay),
                      clip_gradient_norm=kw.pop("clip_gradient_norm", FLAGS.clip_gradient_norm),
                      regularization_penalty=kw.pop("regularization_penalty", FLAGS.regularization_penalty), **kw)
