"""Video-level classifier heads on [B, D_in] features; class names, flags and TF variable names mirror
W/video_level_models.py + W/all_video_models/ (W = /root/reference/youtube-8m-wangheda).

create_model(model_input, vocab_size, **kw) -> {"predictions": [B, V] probabilities, ...}; tensors are
torch tensors on the MI355X and all arithmetic runs in libyt8m_hip.so (ops.py).
"""
import torch

from . import models, ops
from .flags import FLAGS, DEFINE_integer, DEFINE_string, DEFINE_bool
from .variables import get_default_graph, xavier_uniform, zeros

# W/video_level_models.py:19-47
DEFINE_integer("moe_num_mixtures", 2, "The number of mixtures (excluding the dummy 'expert') used for MoeModel.")
DEFINE_integer("deep_chain_layers", 3, "The number of layers used for DeepChainModel")
DEFINE_integer("deep_chain_relu_cells", 200, "The number of relu cells used for DeepChainModel")
DEFINE_string("deep_chain_relu_type", "relu", "The type of relu cells used for DeepChainModel (options are elu and relu)")
DEFINE_bool("deep_chain_use_length", False, "unused by DeepCombineChainModel (kept for flag compatibility)")
# new: MoeModel may return its own "loss" (W/train.py:384-385 honours it) computed by the fused mixing+cross-entropy pass
DEFINE_bool("fused_head_loss", True, "MoeModel returns {'loss': CrossEntropyLoss(predictions, labels)} from a fused kernel "
            "when labels are given, --label_loss=CrossEntropyLoss, no label smoothing and no --multitask.")


def fully_connected(x, num_outputs, scope, activation=None, use_bias=True, l2_penalty=0.0):
    """slim.fully_connected: variables <scope>/weights [in, out] (xavier) and <scope>/biases (zeros)."""
    g = get_default_graph()
    W = g.get_variable(scope + "/weights", (x.shape[-1], num_outputs), xavier_uniform, l2=l2_penalty)
    b = g.get_variable(scope + "/biases", (num_outputs,), zeros) if use_bias else None
    y = ops.linear(x, W, b)
    return ops.activation(y, activation) if activation else y


def fully_connected_cat(parts, num_outputs, scope, activation=None, use_bias=True, l2_penalty=0.0, group_parts=()):
    """slim.fully_connected(tf.concat(parts + tiled group_parts, axis=-1), ...) without materialising the concatenation
    (same variables): group_parts are per-video vectors [B, K] that the reference tiles over the frame axis."""
    g = get_default_graph()
    width = sum(p.shape[-1] for p in parts) + sum(p.shape[-1] for p in group_parts)
    W = g.get_variable(scope + "/weights", (width, num_outputs), xavier_uniform, l2=l2_penalty)
    b = g.get_variable(scope + "/biases", (num_outputs,), zeros) if use_bias else None
    y = ops.linear_cat(list(parts), W, b, group_parts=tuple(group_parts))
    return ops.activation(y, activation) if activation else y


def moe_block(model_input, vocab_size, num_mixtures, l2_penalty, gate_scope, expert_scope, frozen_cols=0):
    """The MoE block shared by MoeModel, the chain models' sub_model and the attention model's sub_moe
    (W/all_video_models/moe_model.py:40-64).  Gate FC has no bias; column l*(M+1)+m = gate m of label l.
    frozen_cols: the leading columns of model_input that are data (ops.moe_head dx_from)."""
    g = get_default_graph()
    d_in = model_input.shape[-1]
    M = num_mixtures
    Wg = g.get_variable(gate_scope + "/weights", (d_in, vocab_size * (M + 1)), xavier_uniform, l2=l2_penalty)
    We = g.get_variable(expert_scope + "/weights", (d_in, vocab_size * M), xavier_uniform, l2=l2_penalty)
    be = g.get_variable(expert_scope + "/biases", (vocab_size * M,), zeros)
    lead = model_input.shape[:-1]
    p = ops.moe_head(model_input.reshape(-1, d_in), Wg, We, be, vocab_size, M, bf16=FLAGS.compute_dtype == "bfloat16", dx_from=frozen_cols)
    return p.view(-1, vocab_size) if len(lead) <= 1 else p.view(-1, vocab_size)


class LogisticModel(models.BaseModel):
    """W/all_video_models/logistic_model.py:9-26: sigmoid(x.W + b), L2 1e-8 on W; scope "fully_connected"."""

    def create_model(self, model_input, vocab_size, l2_penalty=1e-8, original_input=None, **unused_params):
        output = fully_connected(model_input, vocab_size, "fully_connected", activation="sigmoid", l2_penalty=l2_penalty)
        return {"predictions": output}


class MoeModel(models.BaseModel):
    """W/all_video_models/moe_model.py:9-65: per-class softmax over (num_mixtures + 1) logistic experts."""

    def create_model(self, model_input, vocab_size, num_mixtures=None, l2_penalty=1e-8, sub_scope="",
                     original_input=None, labels=None, fuse_loss=True, **unused_params):
        num_mixtures = num_mixtures or FLAGS.moe_num_mixtures
        fused = (labels is not None and fuse_loss and FLAGS.fused_head_loss and FLAGS.label_loss == "CrossEntropyLoss"
                 and not FLAGS.label_smoothing and not FLAGS.multitask and torch.is_grad_enabled()
                 and model_input.dim() == 2 and tuple(labels.shape) == (model_input.shape[0], vocab_size))
        if fused:
            g = get_default_graph()
            d_in, M = model_input.shape[-1], num_mixtures
            Wg = g.get_variable("gates" + sub_scope + "/weights", (d_in, vocab_size * (M + 1)), xavier_uniform, l2=l2_penalty)
            We = g.get_variable("experts" + sub_scope + "/weights", (d_in, vocab_size * M), xavier_uniform, l2=l2_penalty)
            be = g.get_variable("experts" + sub_scope + "/biases", (vocab_size * M,), zeros)
            p, loss = ops.moe_head_xent(model_input, Wg, We, be, labels, vocab_size, M, bf16=FLAGS.compute_dtype == "bfloat16")
            return {"predictions": p, "loss": loss}
        p = moe_block(model_input, vocab_size, num_mixtures, l2_penalty, "gates" + sub_scope, "experts" + sub_scope)
        return {"predictions": p}


class DeepCombineChainModel(models.BaseModel):
    """W/all_video_models/deep_combine_chain_model.py:9-85: chain of MoE sub-predictions, each projected to
    relu cells, L2-normalised and concatenated to the input of the next stage."""

    def create_model(self, model_input, vocab_size, num_mixtures=None, l2_penalty=1e-8, sub_scope="",
                     original_input=None, dropout=False, keep_prob=None, noise_level=None, num_frames=None,
                     support_pool=None, **unused_params):
        """support_pool (this build's addition, None = the reference's behaviour): a callable applied to every stage's
        sub-prediction BEFORE the concatenation into "support_predictions" -- a caller that reduces the rows anyway (the attention
        composite takes the max over its A attention rows per video) then concatenates [B, V] pieces instead of [B * A, V] ones: the
        [B * A, L * V] copy (464 MB at B * A = 8192, L = 3) and the strided gradient slices it leaves behind disappear."""
        num_layers = FLAGS.deep_chain_layers
        relu_cells = FLAGS.deep_chain_relu_cells
        relu_type = FLAGS.deep_chain_relu_type
        next_input = model_input
        # the model input stays in front of every later stage's input (:66-70): when it is data, no head computes a gradient for it
        frozen = 0 if model_input.requires_grad else int(model_input.shape[1])
        support_predictions = []
        for layer in range(num_layers):
            sub_prediction = self.sub_model(next_input, vocab_size, sub_scope=sub_scope + "prediction-%d" % layer,
                                            dropout=dropout, keep_prob=keep_prob, noise_level=noise_level, frozen_cols=frozen)
            sub_activation = fully_connected(sub_prediction, relu_cells, sub_scope + "relu-%d" % layer, l2_penalty=l2_penalty)
            sub_relu = ops.activation(sub_activation, "elu" if relu_type == "elu" else "relu")
            if noise_level is not None:
                sub_relu = ops.add_noise(sub_relu, noise_level)
            relu_norm = ops.l2_normalize(sub_relu)
            next_input = torch.cat([next_input, relu_norm], dim=1)
            support_predictions.append(sub_prediction if support_pool is None else support_pool(sub_prediction))
        main_predictions = self.sub_model(next_input, vocab_size, sub_scope=sub_scope + "-main", frozen_cols=frozen)
        return {"predictions": main_predictions, "support_predictions": torch.cat(support_predictions, dim=1)}

    def sub_model(self, model_input, vocab_size, num_mixtures=None, l2_penalty=1e-8, sub_scope="", dropout=False,
                  keep_prob=None, noise_level=None, frozen_cols=0, **unused_params):
        num_mixtures = num_mixtures or FLAGS.moe_num_mixtures
        if dropout:                                                 # :57-58 tf.nn.dropout on the (grown) chain input
            model_input = ops.dropout(model_input, 1.0 if keep_prob is None else keep_prob)
        return moe_block(model_input, vocab_size, num_mixtures, l2_penalty, "gates-" + sub_scope, "experts-" + sub_scope,
                         frozen_cols=frozen_cols)
