"""Frame-level models on [B, F<=300, D] + num_frames; class names, flags and TF variable names mirror
W/frame_level_models.py + W/all_frame_models/ (W = /root/reference/youtube-8m-wangheda).
NetVLADModel / GatedNetVLADModel are NOT in the reference; they follow SURVEY.md Appendix B.
"""
import math

import torch

from . import models, model_utils, ops, seq_ops, video_level_models
from .flags import FLAGS, DEFINE_integer, DEFINE_bool, DEFINE_string
from .variables import get_default_graph, xavier_uniform, zeros, ones, random_normal

# W/frame_level_models.py:20-84 (hot-path subset)
DEFINE_integer("iterations", 30, "Number of frames per batch for DBoF.")
DEFINE_bool("dbof_add_batch_norm", True, "Adds batch normalization to the DBoF model.")
DEFINE_bool("sample_random_frames", True, "If true samples random frames (for frame level models). If false, a random"
            "sequence of frames is sampled instead.")
DEFINE_integer("dbof_cluster_size", 8192, "Number of units in the DBoF cluster layer.")
DEFINE_integer("dbof_hidden_size", 1024, "Number of units in the DBoF hidden layer.")
DEFINE_string("dbof_pooling_method", "max", "The pooling method used in the DBoF cluster layer. Choices are 'average' and 'max'.")
DEFINE_string("video_level_classifier_model", "MoeModel", "Some Frame-Level models can be decomposed into a "
              "generalized pooling operation followed by a classifier layer")
DEFINE_bool("rnn_swap_memory", False, "If true, swap_memory = True.  (No numerical effect; ignored: 288 GB HBM.)")
DEFINE_string("lstm_cells", "1024", "Number of LSTM cells.")
DEFINE_integer("lstm_layers", 2, "Number of LSTM layers.")
# new: time chunks of the layer-pipelined LSTM stack (1 = one layer after the other)
DEFINE_integer("gru_cells", 1024, "Number of GRU cells.")
DEFINE_integer("gru_layers", 2, "Number of GRU layers.")
DEFINE_integer("lstm_pipeline_chunks", 2, "Time chunks over which the layers of the LSTM stack are pipelined on separate streams.")
DEFINE_string("feature_sizes", "1024", "Length of the feature vectors.")     # W/train.py:58 (read by the parallel LSTM model)
DEFINE_integer("positional_embedding_size", 32, "Positional embedding dimension use in lstm_positional_attention_max_pooling_model.")
DEFINE_integer("lstm_attentions", 8, "Attention size in lstm_attention_max_pooling_model.")
DEFINE_bool("is_training", False, "used in batch normalization.")
# new (Appendix B)
DEFINE_integer("netvlad_cluster_size", 64, "Number of NetVLAD clusters.")
DEFINE_integer("netvlad_hidden_size", 1024, "Width of the FC after the VLAD descriptor.")
DEFINE_bool("netvlad_gating", False, "Context gating after the hidden FC.")
DEFINE_bool("netvlad_add_batch_norm", False, "Kept False so examples stay independent under data parallelism.")


def _head(name=None):
    return getattr(video_level_models, name or FLAGS.video_level_classifier_model)


def _lib_u8_ok(D):
    from . import _lib
    return bool(_lib.lib().yt8m_u8_proj_supported(int(D)))


def _lstm_stack(model_input, num_frames, lstm_size, number_of_layers, scope="RNN", input_keep_prob=None):
    """MultiRNNCell([BasicLSTMCell(H, forget_bias=1.0)] * L) under tf.nn.dynamic_rnn inside variable_scope("RNN")
    (W/all_frame_models/lstm_model.py:34-47).  TF-1.0 variable names:
    RNN/multi_rnn_cell/cell_<l>/basic_lstm_cell/{weights,biases}.  Returns time-major outputs of the top layer
    and the per-layer final (c, h)."""
    g = get_default_graph()
    if model_input.dtype == torch.uint8:
        # raw reader bytes: the stack's layer-0 projection consumes them directly (csrc/u8proj.hip: exact bf16 operands, the
        # dequantise / l2-normalise affine folded into the GEMM epilogue); no fp32 [B,F,D] tensor, no transpose copy
        dropping = input_keep_prob is not None and float(input_keep_prob) < 1.0
        if _lib_u8_ok(model_input.shape[2]) and not dropping:
            x_tm = model_input                                   # [B,F,D] uint8, re-ordered time-major by the conversion pass
        else:
            x_tm = ops.dequant_l2norm(model_input, num_frames).transpose(0, 1).contiguous()
    else:
        x_tm = model_input.transpose(0, 1).contiguous()          # [F,B,D]   (layout glue)
    wb = []
    d_in = model_input.shape[2]
    with g.variable_scope(scope):
        for l in range(number_of_layers):
            scope = "multi_rnn_cell/cell_%d/basic_lstm_cell" % l
            W = g.get_variable(scope + "/weights", (d_in + lstm_size, 4 * lstm_size), xavier_uniform)
            b = g.get_variable(scope + "/biases", (4 * lstm_size,), zeros)
            wb.append((W, b))
            d_in = lstm_size
    # all layers in one op: layer l+1 works on time chunk c while layer l is already in chunk c+1 (seq_ops._LstmStack)
    return seq_ops.lstm_stack(x_tm, num_frames, wb, forget_bias=1.0, chunks=FLAGS.lstm_pipeline_chunks,
                              input_keep_prob=input_keep_prob, bf16=FLAGS.compute_dtype == "bfloat16")


class FrameLevelLogisticModel(models.BaseModel):
    """W/all_frame_models/logistic_model.py:13-46: logistic classifier over the num_frames-average of the frames."""

    accepts_quantized_input = True

    def create_model(self, model_input, vocab_size, num_frames, **unused_params):
        if model_input.dtype == torch.uint8 and seq_ops.u8_attention_supported(model_input, 1):
            # the reader's bytes: sum_f x[b, f] / num_frames as one weighted pooling pass over them (rs is 0 on the padding frames)
            B, F, D = model_input.shape
            q = model_input.contiguous()
            rs = seq_ops.u8_frame_scales(q, num_frames)
            inv = 1.0 / num_frames.to(torch.float32)
            avg_pooled = seq_ops.pool_u8_raw(inv.view(B, 1, 1).expand(B, F, 1).contiguous(), q, rs).view(B, D)
        else:
            if model_input.dtype == torch.uint8:
                model_input = ops.dequant_l2norm(model_input, num_frames)
            denominators = num_frames.to(torch.float32).unsqueeze(1)
            avg_pooled = model_input.sum(dim=1) / denominators     # input is data: no gradient flows here
        output = video_level_models.fully_connected(avg_pooled, vocab_size, "fully_connected", activation="sigmoid",
                                                    l2_penalty=1e-8)
        return {"predictions": output}


class LstmModel(models.BaseModel):
    """W/all_frame_models/lstm_model.py:13-57: the head reads the whole final state [c0||h0||c1||h1]
    (state_is_tuple=False), 4H wide for two layers.  accepts_quantized_input: see _lstm_stack."""
    accepts_quantized_input = True

    def create_model(self, model_input, vocab_size, num_frames, **unused_params):
        lstm_size = int(FLAGS.lstm_cells)
        number_of_layers = FLAGS.lstm_layers
        _, finals = _lstm_stack(model_input, num_frames, lstm_size, number_of_layers)
        state = torch.cat([t for pair in finals for t in pair], dim=1)
        return _head()().create_model(model_input=state, original_input=model_input, vocab_size=vocab_size,
                                      **unused_params)


class LstmMemoryModel(models.BaseModel):
    """W/all_frame_models/lstm_memory_model.py:13-73: the head reads the concatenated c states (2H)."""
    accepts_quantized_input = True

    def create_model(self, model_input, vocab_size, num_frames, dropout=False, keep_prob=None, noise_level=None,
                     **unused_params):
        lstm_size = int(FLAGS.lstm_cells)
        number_of_layers = FLAGS.lstm_layers
        # :36-45 DropoutWrapper(BasicLSTMCell, input_keep_prob=keep_prob) around every layer when --dropout
        _, finals = _lstm_stack(model_input, num_frames, lstm_size, number_of_layers,
                                input_keep_prob=keep_prob if dropout else None)
        final_state = torch.cat([c for c, _ in finals], dim=1)
        if noise_level is not None:
            final_state = ops.add_noise(final_state, noise_level)
        return _head()().create_model(model_input=final_state, original_input=model_input, vocab_size=vocab_size,
                                      num_frames=num_frames, **unused_params)


def _recurrent_input(model_input, num_frames):
    """Layer-0 input of a GRU / LayerNorm-LSTM stack: the reader's bytes as operand images (seq_ops.U8FrameImages: the hoisted input
    projection and its weight gradient read them, no fp32 [B,F,D] tensor) where the byte products cover the shape, else the float
    frames time-major [F,B,D] (dequantised first when they arrive as bytes)."""
    if model_input.dtype == torch.uint8:
        if seq_ops.u8_hoisted_supported(model_input):
            return seq_ops.U8FrameImages(model_input, num_frames)
        model_input = ops.dequant_l2norm(model_input, num_frames)
    return model_input.transpose(0, 1).contiguous()          # (layout glue)


def _gru_stack(model_input, num_frames, gru_size, number_of_layers):
    """MultiRNNCell([GRUCell(H)] * L, state_is_tuple=False) under tf.nn.dynamic_rnn in variable_scope("RNN")
    (W/all_frame_models/gru_pooling_model.py:34-47).  TF-1.0 names: RNN/multi_rnn_cell/cell_<l>/gru_cell/{gates,candidate}/
    {weights,biases}; the gate bias starts at 1.  Returns (top outputs time-major [F,B,H], [h_l final])."""
    g = get_default_graph()
    x_tm = _recurrent_input(model_input, num_frames)
    finals = []
    d_in = model_input.shape[2]
    with g.variable_scope("RNN"):
        for l in range(number_of_layers):
            scope = "multi_rnn_cell/cell_%d/gru_cell" % l
            Wg = g.get_variable(scope + "/gates/weights", (d_in + gru_size, 2 * gru_size), xavier_uniform)
            bg = g.get_variable(scope + "/gates/biases", (2 * gru_size,), ones)
            Wc = g.get_variable(scope + "/candidate/weights", (d_in + gru_size, gru_size), xavier_uniform)
            bc = g.get_variable(scope + "/candidate/biases", (gru_size,), zeros)
            x_tm, h = seq_ops.gru_layer(x_tm, Wg, bg, Wc, bc, num_frames)
            finals.append(h)
            d_in = gru_size
    return x_tm, finals


def _mean_over_frames(out_tm, num_frames):
    """reduce_sum(outputs, axis=1) / max(num_frames, 1) (W/all_frame_models/gru_pooling_model.py:48-49): dynamic_rnn outputs
    are zero past num_frames, so this is one [1,F] x [F,H] product per video with constant weights 1 / max(num_frames, 1)."""
    F, B, H = out_tm.shape
    w = (1.0 / num_frames.to(torch.float32).clamp(min=1.0)).view(B, 1, 1).expand(B, F, 1).contiguous()
    return seq_ops.pool_tn(w, out_tm.transpose(0, 1).contiguous()).view(B, H)


class GruPoolingModel(models.BaseModel):
    """W/all_frame_models/gru_pooling_model.py:13-58: GRU stack, head input = outputs averaged over the video's frames.
    (The reference file divides by tf.maximum(num_frames, tf.ones([batch_size, 1])) with `batch_size` undefined -- it raises
    NameError at graph construction; built here with the evident meaning.)"""
    accepts_quantized_input = True                         # _recurrent_input: layer 0 reads the reader's bytes

    def create_model(self, model_input, vocab_size, num_frames, **unused_params):
        out_tm, _ = _gru_stack(model_input, num_frames, FLAGS.gru_cells, FLAGS.gru_layers)
        pooling_output = _mean_over_frames(out_tm, num_frames)
        return _head()().create_model(model_input=pooling_output, original_input=model_input, vocab_size=vocab_size,
                                      **unused_params)


class GruWithPoolingModel(models.BaseModel):
    """W/all_frame_models/gru_with_pooling_model.py:13-60: head input = [mean-pooled outputs || final state of every layer]
    (state_is_tuple=False: [h_0 || h_1 ...]).  Same `batch_size` NameError in the reference as GruPoolingModel."""
    accepts_quantized_input = True                         # _recurrent_input: layer 0 reads the reader's bytes

    def create_model(self, model_input, vocab_size, num_frames, **unused_params):
        out_tm, finals = _gru_stack(model_input, num_frames, FLAGS.gru_cells, FLAGS.gru_layers)
        final_output = torch.cat([_mean_over_frames(out_tm, num_frames)] + finals, dim=1)
        return _head()().create_model(model_input=final_output, original_input=model_input, vocab_size=vocab_size,
                                      **unused_params)


LN_GATES = ("input", "transform", "forget", "output", "state")


class LayerNormLstmMemoryModel(models.BaseModel):
    """W/all_frame_models/layernorm_lstm_memory_model.py:13-72: MultiRNNCell([LayerNormBasicLSTMCell(H)] * L); with --dropout the
    cells get dropout_keep_prob=keep_prob (recurrent dropout on the candidate); head input = concat of the (normalised) c
    states.  TF-1.0 names: RNN/multi_rnn_cell/cell_<l>/layer_norm_basic_lstm_cell/{weights, <gate>/gamma, <gate>/beta}."""
    accepts_quantized_input = True                         # _recurrent_input: layer 0 reads the reader's bytes

    def create_model(self, model_input, vocab_size, num_frames, dropout=False, keep_prob=None, noise_level=None,
                     **unused_params):
        lstm_size = int(FLAGS.lstm_cells)
        g = get_default_graph()
        x_tm = _recurrent_input(model_input, num_frames)
        cs = []
        d_in = model_input.shape[2]
        with g.variable_scope("RNN"):
            for l in range(FLAGS.lstm_layers):
                scope = "multi_rnn_cell/cell_%d/layer_norm_basic_lstm_cell" % l
                W = g.get_variable(scope + "/weights", (d_in + lstm_size, 4 * lstm_size), xavier_uniform)
                gammas = [g.get_variable("%s/%s/gamma" % (scope, n), (lstm_size,), ones) for n in LN_GATES]
                betas = [g.get_variable("%s/%s/beta" % (scope, n), (lstm_size,), zeros) for n in LN_GATES]
                x_tm, c, _ = seq_ops.lnlstm_layer(x_tm, W, gammas, betas, num_frames, forget_bias=1.0,
                                                  keep_prob=keep_prob if (dropout and keep_prob is not None) else 1.0)
                cs.append(c)
                d_in = lstm_size
        final_state = torch.cat(cs, dim=1)
        if noise_level is not None:
            final_state = ops.add_noise(final_state, noise_level)
        return _head()().create_model(model_input=final_state, original_input=model_input, vocab_size=vocab_size,
                                      **unused_params)


def _u8_or_float(model_input, num_frames, num_attentions):
    """(input, True) when the raw uint8 frames can go all the way (the stack's layer-0 projection and the attention FC both read bytes);
    else the dequantised, l2-normalised float frames the reference's transformer would have handed over."""
    if model_input.dtype != torch.uint8:
        return model_input, False
    if _lib_u8_ok(model_input.shape[2]) and seq_ops.u8_attention_supported(model_input, num_attentions):
        return model_input.contiguous(), True
    return ops.dequant_l2norm(model_input, num_frames), False


def _attention_fc_u8(q, num_frames, parts, num_outputs, scope, l2_penalty, rs=None):
    """slim.fully_connected(concat([x] + parts)) with x = the raw frames (same variables as video_level_models.fully_connected_cat);
    parts: [B,F,K] per-frame tensors and [B,K] per-video vectors (tiled over the frames by the reference), in concatenation order."""
    g = get_default_graph()
    width = q.shape[2] + sum(p.shape[-1] for p in parts)
    W = g.get_variable(scope + "/weights", (width, num_outputs), xavier_uniform, l2=l2_penalty)
    b = g.get_variable(scope + "/biases", (num_outputs,), zeros)
    if rs is None:
        rs = seq_ops.u8_frame_scales(q, num_frames)                                 # [B,F]; 0 on the padding frames
    return seq_ops.attention_logits_u8(q, rs, None, W, b, parts=parts)


class LstmAttentionMaxPoolingModel(models.BaseModel):
    """W/all_frame_models/lstm_attention_max_pooling_model.py:10-98: LSTM outputs -> A attention poolings ->
    MoE per attention -> max over attentions.  accepts_quantized_input: the stack (see _lstm_stack) and the attention FC read the raw
    reader bytes; no fp32 [B,F,D] tensor."""
    accepts_quantized_input = True

    def create_model(self, model_input, vocab_size, num_frames, num_mixtures=None, l2_penalty=1e-8, sub_scope="",
                     original_input=None, **unused_params):
        lstm_size = int(FLAGS.lstm_cells)
        number_of_layers = FLAGS.lstm_layers
        num_attentions = FLAGS.lstm_attentions
        model_input, u8 = _u8_or_float(model_input, num_frames, num_attentions)
        out_tm, _ = _lstm_stack(model_input, num_frames, lstm_size, number_of_layers)
        outputs = out_tm.transpose(0, 1).contiguous()                               # [B,F,H]
        if u8:                                                                      # raw reader bytes into the stack AND the attention FC
            attention_activations = _attention_fc_u8(model_input, num_frames, [outputs], num_attentions, "attention-" + sub_scope,
                                                     l2_penalty)
        else:
            attention_activations = video_level_models.fully_connected_cat(           # :51-56 FC on concat([input, outputs])
                [model_input, outputs], num_attentions, "attention-" + sub_scope, l2_penalty=l2_penalty)
        attention_weights = seq_ops.attention_weights(attention_activations, num_frames)   # [B,F,A]
        attention_outputs = seq_ops.pool_tn(attention_weights, outputs)                    # [B,A,H]
        moe_predictions = self.sub_moe(attention_outputs, vocab_size, sub_scope="sub-moe")
        predictions = moe_predictions.view(-1, num_attentions, vocab_size)
        max_predictions = ops.frame_pool(predictions, "max")          # tf.reduce_max over the attentions
        return {"predictions": max_predictions}

    def sub_moe(self, model_input, vocab_size, num_mixtures=None, l2_penalty=1e-8, sub_scope="", **unused_params):
        num_mixtures = num_mixtures or FLAGS.moe_num_mixtures
        return video_level_models.moe_block(model_input, vocab_size, num_mixtures, l2_penalty,
                                            "gates-" + sub_scope, "experts-" + sub_scope)


class LstmParallelFinaloutputModel(models.BaseModel):
    """W/all_frame_models/lstm_parallel_finaloutput_model.py:13-73: one LSTM stack per input feature (rgb / audio: the
    input is split by --feature_sizes, each part re-normalised), head input = concat of every layer's final h.
    accepts_quantized_input: l2_normalize(slice of l2_normalize(x)) = l2_normalize(slice of x), so a stack whose slice the uint8
    projection covers reads the reader's bytes of that slice (its own row norms folded into the GEMM epilogue, see _lstm_stack); the
    other slices are dequantised and normalised as floats."""
    accepts_quantized_input = True

    def create_model(self, model_input, vocab_size, num_frames, **unused_params):
        number_of_layers = FLAGS.lstm_layers
        lstm_sizes = [int(v) for v in str(FLAGS.lstm_cells).split(",")]
        feature_sizes = [int(v) for v in str(FLAGS.feature_sizes).split(",")]
        assert len(lstm_sizes) == len(feature_sizes), \
            "length of lstm_sizes (={}) != length of feature_sizes (={})".format(len(lstm_sizes), len(feature_sizes))
        assert sum(feature_sizes) == model_input.shape[2], "feature_sizes do not add up to the input width"
        states, off = [], 0
        for i, (fs, hs) in enumerate(zip(feature_sizes, lstm_sizes)):
            sub_input = model_input[:, :, off:off + fs].contiguous()
            if sub_input.dtype != torch.uint8:
                sub_input = ops.l2_normalize(sub_input)
            elif not _lib_u8_ok(fs):                                  # (dequant_l2norm normalises the slice and zeroes the padding frames)
                sub_input = ops.dequant_l2norm(sub_input, num_frames)
            off += fs
            _, finals = _lstm_stack(sub_input, num_frames, hs, number_of_layers, scope="RNN%d" % i)
            states.extend(h for _, h in finals)
        final_state = torch.cat(states, dim=1)
        return _head()().create_model(model_input=final_state, original_input=model_input, vocab_size=vocab_size,
                                      **unused_params)


class LstmPositionalAttentionMaxPoolingModel(LstmAttentionMaxPoolingModel):
    """W/all_frame_models/lstm_positional_attention_max_pooling_model.py:10-87: as LstmAttentionMaxPoolingModel, the
    attention FC additionally sees a learned positional embedding [1,F,E] and the masked mean of the input."""

    def create_model(self, model_input, vocab_size, num_frames, num_mixtures=None, l2_penalty=1e-8, sub_scope="",
                     original_input=None, **unused_params):
        lstm_size = int(FLAGS.lstm_cells)
        num_attentions = FLAGS.lstm_attentions
        B, F, D = model_input.shape
        model_input, u8 = _u8_or_float(model_input, num_frames, num_attentions)
        out_tm, _ = _lstm_stack(model_input, num_frames, lstm_size, FLAGS.lstm_layers)
        outputs = out_tm.transpose(0, 1).contiguous()                               # [B,F,H]
        g = get_default_graph()
        emb = g.get_variable("positional_embedding", (1, F, FLAGS.positional_embedding_size), xavier_uniform, l2=l2_penalty)
        positional_embedding = ops.as_tensor(emb).expand(B, F, FLAGS.positional_embedding_size)
        if u8:
            # raw reader bytes: the masked mean frame from the bytes (rs is 0 on the padding frames), the FC on [x | emb | mean | outputs]
            # with x read as bytes and the mean as ONE row per video
            rs = seq_ops.u8_frame_scales(model_input, num_frames)
            inv = 1.0 / num_frames.to(torch.float32)
            mean_input = seq_ops.pool_u8_raw(inv.view(B, 1, 1).expand(B, F, 1).contiguous(), model_input, rs).view(B, D)
            attention_activations = _attention_fc_u8(model_input, num_frames, [positional_embedding.contiguous(), mean_input, outputs],
                                                     num_attentions, "attention-" + sub_scope, l2_penalty, rs=rs)
        else:
            mask = (torch.arange(F, device=model_input.device)[None, :] < num_frames[:, None]).to(model_input.dtype)
            mean_input = (model_input * mask[:, :, None]).sum(dim=1) / num_frames.to(model_input.dtype)[:, None]
            attention_activations = video_level_models.fully_connected_cat(
                [model_input, positional_embedding, mean_input[:, None, :].expand(B, F, D), outputs],
                num_attentions, "attention-" + sub_scope, l2_penalty=l2_penalty)
        attention_weights = seq_ops.attention_weights(attention_activations, num_frames)   # [B,F,A]
        attention_outputs = seq_ops.pool_tn(attention_weights, outputs)                    # [B,A,H]
        moe_predictions = self.sub_moe(attention_outputs, vocab_size, sub_scope="sub-moe")
        predictions = moe_predictions.view(-1, num_attentions, vocab_size)
        return {"predictions": ops.frame_pool(predictions, "max")}


class CnnDeepCombineChainModel(models.BaseModel):
    """W/all_frame_models/cnn_deep_combine_chain_model.py:10-140: chain of MoE sub-predictions whose inputs are max-pooled
    "einsum CNNs" over the frames (filter lengths 1,2,3 = GEMMs on the input concatenated with its 1- and 2-frame shifts),
    the masked mean input and the l2-normalised relu projections of the previous predictions.
    accepts_quantized_input: on the reader's bytes every CNN of the chain reads ONE half image of the frames (seq_ops.u8_cnn: a shift by
    i frames is a row offset in time-major order -- no concatenated [B,F,2D] / [B,F,3D] tensors, no fp32 copy of the frames), the mean
    frame comes from the bytes too."""
    accepts_quantized_input = True

    def cnn(self, model_input, l2_penalty=1e-8, num_filters=(1024, 1024, 1024), filter_sizes=(1, 2, 3), sub_scope="",
            **unused_params):
        g = get_default_graph()
        B, F, D = model_input.shape
        shift_inputs = [model_input]
        for i in range(1, max(filter_sizes)):                         # tf.pad(..., [[0,0],[i,0],[0,0]])[:, :F]
            shift_inputs.append(torch.cat([model_input.new_zeros(B, i, D), model_input[:, :F - i]], dim=1))
        cnn_outputs = []
        for nf, fs in zip(num_filters, filter_sizes):
            sub_input = torch.cat(shift_inputs[:fs], dim=2) if fs > 1 else shift_inputs[0]
            sub_filter = g.get_variable(sub_scope + "cnn-filter-len%d" % fs, (D * fs, nf), random_normal(0.1), l2=l2_penalty)
            cnn_outputs.append(ops.linear(sub_input, sub_filter))
        return torch.cat(cnn_outputs, dim=2)

    def create_model(self, model_input, vocab_size, num_frames, num_mixtures=None, l2_penalty=1e-8, sub_scope="",
                     original_input=None, **unused_params):
        num_layers = FLAGS.deep_chain_layers
        relu_cells = FLAGS.deep_chain_relu_cells
        B, F, D = model_input.shape
        frames = None
        if model_input.dtype == torch.uint8:
            if seq_ops.u8_cnn_supported(model_input) and seq_ops.u8_attention_supported(model_input, 1):
                frames = seq_ops.U8FrameImages(model_input, num_frames)         # one byte image for every CNN of the chain
            else:
                model_input = ops.dequant_l2norm(model_input, num_frames)
        if frames is not None:
            rs = seq_ops.u8_frame_scales(frames.q, num_frames)                  # [B,F]; 0 on the padding frames
            inv = 1.0 / num_frames.to(torch.float32)
            mean_input = seq_ops.pool_u8_raw(inv.view(B, 1, 1).expand(B, F, 1).contiguous(), frames.q, rs).view(B, D)
        else:
            mask = (torch.arange(F, device=model_input.device)[None, :] < num_frames[:, None]).to(model_input.dtype)
            mean_input = (model_input * mask[:, :, None]).sum(dim=1) / num_frames.to(model_input.dtype)[:, None]
        mean_relu = video_level_models.fully_connected(mean_input, relu_cells, sub_scope + "mean-relu", activation="relu",
                                                       l2_penalty=l2_penalty)
        relu_layers = [ops.l2_normalize(mean_relu)]
        filters = dict(num_filters=[relu_cells, relu_cells, relu_cells * 2], filter_sizes=[1, 2, 3])

        def pooled_cnn(scope):
            if frames is not None:
                g = get_default_graph()
                fvars = [g.get_variable(scope + "cnn-filter-len%d" % fs, (D * fs, nfl), random_normal(0.1), l2=l2_penalty)
                         for nfl, fs in zip(filters["num_filters"], filters["filter_sizes"])]
                if sum(filters["num_filters"]) % 4 == 0:                          # pooled in time-major order, sparse weight gradient
                    return ops.l2_normalize(seq_ops.u8_cnn_maxpool(frames, fvars))
                cnn_output = seq_ops.u8_cnn(frames, fvars)
            else:
                cnn_output = self.cnn(model_input, sub_scope=scope, l2_penalty=l2_penalty, **filters)
            return ops.l2_normalize(ops.frame_pool(cnn_output, "max"))      # reduce_max over ALL max_frames rows, as the reference

        next_input = pooled_cnn(sub_scope + "cnn0")
        frozen = 0
        support_predictions = []
        for layer in range(num_layers):
            sub_prediction = self.sub_model(next_input, vocab_size, sub_scope=sub_scope + "prediction-%d" % layer, frozen_cols=frozen)
            support_predictions.append(sub_prediction)
            sub_relu = video_level_models.fully_connected(sub_prediction, relu_cells, sub_scope + "relu-%d" % layer,
                                                          activation="relu", l2_penalty=l2_penalty)
            relu_layers.append(ops.l2_normalize(sub_relu))
            normalized_cnn_output = pooled_cnn(sub_scope + "cnn%d" % (layer + 1))
            next_input = torch.cat([mean_input, normalized_cnn_output] + relu_layers, dim=1)
            frozen = 0 if mean_input.requires_grad else D           # the mean frame in front of the stage's input is data
        main_predictions = self.sub_model(next_input, vocab_size, sub_scope=sub_scope + "-main", frozen_cols=frozen)
        return {"predictions": main_predictions, "support_predictions": torch.cat(support_predictions, dim=1)}

    def sub_model(self, model_input, vocab_size, num_mixtures=None, l2_penalty=1e-8, sub_scope="", frozen_cols=0, **unused_params):
        num_mixtures = num_mixtures or FLAGS.moe_num_mixtures
        return video_level_models.moe_block(model_input, vocab_size, num_mixtures, l2_penalty,
                                            "gates-" + sub_scope, "experts-" + sub_scope, frozen_cols=frozen_cols)


def _batch_norm(x, scope, is_training, eps=1e-3, decay=0.999):
    """slim.batch_norm(center=True, scale=True): batch moments, moving averages and the backward all in
    csrc/dbof.hip (ops.batch_norm).  Couples the examples of the local batch, like the reference."""
    g = get_default_graph()
    n = x.shape[-1]
    gamma = g.get_variable(scope + "/gamma", (n,), ones)
    beta = g.get_variable(scope + "/beta", (n,), zeros)
    mm = g.get_variable(scope + "/moving_mean", (n,), zeros, trainable=False)
    mv = g.get_variable(scope + "/moving_variance", (n,), ones, trainable=False)
    return ops.batch_norm(x, gamma, beta, mm, mv, is_training, eps, decay)


class DbofModel(models.BaseModel):
    """W/all_frame_models/dbof_model.py:13-124: sample frames -> cluster FC -> (BN) -> relu6 -> pool over frames ->
    hidden FC -> (BN) -> relu6 -> head.  Weight variables are anonymous tf.Variable()s in the reference.

    accepts_quantized_input: raw uint8 frames are sampled FIRST (csrc/dbof.hip) and only the `iterations` sampled frames per
    video are dequantised + l2-normalised -- 30 of 300 rows; the fp32 [B,300,1152] tensor is never written."""
    accepts_quantized_input = True

    def create_model(self, model_input, vocab_size, num_frames, iterations=None, add_batch_norm=None,
                     sample_random_frames=None, cluster_size=None, hidden_size=None, is_training=True,
                     **unused_params):
        iterations = iterations or FLAGS.iterations
        add_batch_norm = add_batch_norm or FLAGS.dbof_add_batch_norm
        random_frames = sample_random_frames or FLAGS.sample_random_frames
        cluster_size = cluster_size or FLAGS.dbof_cluster_size
        hidden1_size = hidden_size or FLAGS.dbof_hidden_size
        g = get_default_graph()
        if random_frames:
            model_input = model_utils.SampleRandomFrames(model_input, num_frames, iterations)
        else:
            model_input = model_utils.SampleRandomSequence(model_input, num_frames, iterations)
        if model_input.dtype == torch.uint8:        # sampled frames are valid ones (all S, or none for a video without frames)
            nf_s = None if num_frames is None else (num_frames.to(torch.int32) > 0).to(torch.int32) * iterations
            model_input = ops.dequant_l2norm(model_input, nf_s)
        max_frames, feature_size = model_input.shape[1], model_input.shape[2]
        reshaped_input = model_input.reshape(-1, feature_size)
        if add_batch_norm:
            reshaped_input = _batch_norm(reshaped_input, "input_bn", is_training)
        cluster_weights = g.anonymous_variable((feature_size, cluster_size), random_normal(1 / math.sqrt(feature_size)))
        if add_batch_norm:
            activation = ops.linear(reshaped_input, cluster_weights)
            activation = _batch_norm(activation, "cluster_bn", is_training)
        else:
            cluster_biases = g.anonymous_variable((cluster_size,), random_normal(1 / math.sqrt(feature_size)))
            activation = ops.linear(reshaped_input, cluster_weights, cluster_biases)
        activation = ops.activation(activation, "relu6")
        activation = activation.view(-1, max_frames, cluster_size)
        activation = model_utils.FramePooling(activation, FLAGS.dbof_pooling_method)
        hidden1_weights = g.anonymous_variable((cluster_size, hidden1_size), random_normal(1 / math.sqrt(cluster_size)))
        if add_batch_norm:
            activation = ops.linear(activation, hidden1_weights)
            activation = _batch_norm(activation, "hidden1_bn", is_training)
        else:
            hidden1_biases = g.anonymous_variable((hidden1_size,), random_normal(0.01))
            activation = ops.linear(activation, hidden1_weights, hidden1_biases)
        activation = ops.activation(activation, "relu6")
        return _head()().create_model(model_input=activation, original_input=model_input, vocab_size=vocab_size,
                                      **unused_params)


class NetVLADModel(models.BaseModel):
    """soft-assignment + residual aggregation + intra-norm + L2 + hidden FC (+ context gating).

    accepts_quantized_input: the trainer may hand over the reader's RAW uint8 frames [B,F,D] instead of running the
    DefaultTransformer first; dequantise + l2-normalise are then folded into the pooling GEMMs (csrc/netvlad_fused.hip)
    and the fp32 [B,F,D] tensor is never written.  float inputs (or shapes the fused kernels do not cover) take the
    generic GEMM + softmax + batched-GEMM path; both give the same function."""
    gating = None
    accepts_quantized_input = True

    def descriptor(self, model_input, num_frames, cluster_size=None, hidden_size=None, gating=None):
        """[B,F,D] frames (uint8 raw or float transformed) -> hidden descriptor h [B, netvlad_hidden_size]."""
        K = cluster_size or FLAGS.netvlad_cluster_size
        Hfc = hidden_size or FLAGS.netvlad_hidden_size
        gating = (self.gating if self.gating is not None else FLAGS.netvlad_gating) if gating is None else gating
        g = get_default_graph()
        B, F, D = model_input.shape
        Wc = g.get_variable("netvlad/cluster_weights", (D, K), random_normal(1 / math.sqrt(D)))
        bc = g.get_variable("netvlad/cluster_biases", (K,), zeros)
        centres = g.get_variable("netvlad/centres", (K, D), random_normal(1 / math.sqrt(D)))
        # The descriptor-wide l2-normalisation needs no pass over [B,K,D]: the intra-normalised rows have the squared norms q
        # (1 unless clamped) the finishing kernel hands out, so v = vlad * s with s = rsqrt(max(sum_k q, eps)) per video, and
        # v . W = s * (vlad . W): the scale goes onto the [B, hidden] output (same function and gradients as l2_normalize first).
        want_q = seq_ops.vlad_q_supported(D)
        if model_input.dtype == torch.uint8 and seq_ops.netvlad_fused_supported(model_input, K):
            nsplit = 1 if FLAGS.compute_dtype == "bfloat16" else 2        # f16 operands vs f16 hi+lo (fp32-class)
            vlad = seq_ops.netvlad_pool_u8(model_input, num_frames, Wc, bc, centres, nsplit, want_q=want_q)
        else:
            if model_input.dtype == torch.uint8:                          # shapes outside the fused kernels' cover
                model_input = ops.dequant_l2norm(model_input, num_frames)
            s = ops.linear(model_input, Wc, bc)                           # [B,F,K] assignment logits
            a = seq_ops.masked_softmax_rows(s, num_frames)                # softmax_k * mask
            agg = seq_ops.pool_tn(a, model_input)                         # [B,K,D] = a^T x per video
            vlad = seq_ops.vlad_finish(agg, a, centres, want_q=want_q)    # (agg - n*c), intra-normalised per cluster
        if want_q:
            vlad, qn = vlad
            scale = torch.rsqrt(torch.clamp(qn.sum(dim=1), min=1e-12)).unsqueeze(1)          # [B,1]; eps of ops.l2_normalize
            Wh = g.get_variable("netvlad/hidden/weights", (K * D, Hfc), video_level_models.xavier_uniform)
            bh = g.get_variable("netvlad/hidden/biases", (Hfc,), video_level_models.zeros)
            h = ops.linear(vlad.reshape(B, K * D), Wh, None) * scale + ops.as_tensor(bh)
        else:
            v = ops.l2_normalize(vlad.reshape(B, K * D))
            h = video_level_models.fully_connected(v, Hfc, "netvlad/hidden")
        if gating:
            gate = video_level_models.fully_connected(h, Hfc, "netvlad/gating", activation="sigmoid")
            h = h * gate
        return h

    def create_model(self, model_input, vocab_size, num_frames, cluster_size=None, hidden_size=None, gating=None,
                     **unused_params):
        h = self.descriptor(model_input, num_frames, cluster_size, hidden_size, gating)
        return _head()().create_model(model_input=h, original_input=model_input, vocab_size=vocab_size,
                                      **unused_params)


class GatedNetVLADModel(NetVLADModel):
    gating = True


class GatedNetVLADAttentionChainModel(GatedNetVLADModel):
    """BASELINE configs[4] composite ("Gated-NetVLAD + attention pooling + chained MoE"), fixed in SURVEY.md Appendix B
    from reference parts (NOT a reference class):
      (i)   h[b]       = gated NetVLAD descriptor (GatedNetVLADModel.descriptor);
      (ii)  att[b,a,:] = sum_f w[b,f,a] x[b,f,:],  w = renorm(mask * softmax_F([x || mean_x] W_a + b_a)) -- the attention
            pooling of W/all_frame_models/lstm_attention_max_pooling_model.py:34,51-63 with the LSTM outputs replaced by
            the frames themselves, A = --lstm_attentions;
      (iii) [h[b] || att[b,a]] -> DeepCombineChainModel (W/all_video_models/deep_combine_chain_model.py:12-85) on the
            [B*A] rows -> max over the A attentions (lstm_attention_max_pooling_model.py:64-66), for the predictions AND the
            support predictions so that they line up with MultiTaskCrossEntropyLoss.get_support (W/losses.py:225-255)."""

    def create_model(self, model_input, vocab_size, num_frames, l2_penalty=1e-8, **unused_params):
        A = FLAGS.lstm_attentions
        B, F, D = model_input.shape
        h = self.descriptor(model_input, num_frames)                                       # [B,Hfc] (uint8 stays fused)
        if seq_ops.u8_attention_supported(model_input, A):
            # raw reader bytes all the way: the logit FC, the pooling and their gradients read uint8 (csrc/gemm_skinny.hip)
            q = model_input.contiguous()
            rs = seq_ops.u8_frame_scales(q, num_frames)                                    # [B,F]; 0 on the padding frames
            inv = (1.0 / num_frames.to(torch.float32).clamp(min=1)) if num_frames is not None else \
                torch.full((B,), 1.0 / F, dtype=torch.float32, device=q.device)
            mean_x = seq_ops.pool_u8_raw(inv.view(B, 1, 1).expand(B, F, 1).contiguous(), q, rs).view(B, D)
            g = video_level_models.get_default_graph()
            W = g.get_variable("attention-/weights", (2 * D, A), video_level_models.xavier_uniform, l2=l2_penalty)
            b = g.get_variable("attention-/biases", (A,), video_level_models.zeros)
            act = seq_ops.attention_logits_u8(q, rs, mean_x, W, b)
            w = seq_ops.attention_weights(act, num_frames)                                 # [B,F,A]
            att = seq_ops.pool_tn_u8(w, q, rs)                                             # [B,A,D]
        else:
            x = ops.dequant_l2norm(model_input, num_frames) if model_input.dtype == torch.uint8 else model_input
            nf = num_frames.to(x.dtype).clamp(min=1).view(B, 1, 1) if num_frames is not None else float(F)
            mean_x = (x.sum(dim=1, keepdim=True) / nf).view(B, D)                          # tiled over the frames by the FC
            act = video_level_models.fully_connected_cat([x], A, "attention-", l2_penalty=l2_penalty, group_parts=[mean_x])
            w = seq_ops.attention_weights(act, num_frames)                                 # [B,F,A]
            att = seq_ops.pool_tn(w, x)                                                    # [B,A,D]
        chain_in = torch.cat([h.unsqueeze(1).expand(B, A, h.shape[1]), att], dim=2).reshape(B * A, -1)
        unused_params.pop("original_input", None)
        # max over the A attention rows of a video (lstm_attention_max_pooling_model.py:64-66), stage by stage: the support
        # predictions are pooled to [B, V] BEFORE they are concatenated (max over rows commutes with concatenation along columns)
        pool = lambda p_: ops.frame_pool(p_.view(B, A, vocab_size), "max")
        res = video_level_models.DeepCombineChainModel().create_model(chain_in, vocab_size, l2_penalty=l2_penalty,
                                                                      original_input=model_input, support_pool=pool, **unused_params)
        return {"predictions": pool(res["predictions"]), "support_predictions": res["support_predictions"]}
