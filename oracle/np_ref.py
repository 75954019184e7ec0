"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- numpy restatement of the hot path.

All functions compute in the dtype of their inputs (tests feed float64 for the "truth" and
float32 to mimic the reference's arithmetic).  Citations are relative to
/root/reference/youtube-8m-wangheda/ (W) unless prefixed; "A.n" = SURVEY.md Appendix A item n
(TensorFlow-1.0 semantics: third-party, un-vendored submodule pinned "Tensorflow 1.0" by
README.md:8 and cloudml-gpu.yaml:5).  **Parity unpinned** for everything in this file: the
reference holds no tests / golden vectors for it and cannot run here.
"""
import numpy as np

XENT_EPS = 10e-6  # W/losses.py:115  (== 1e-5, *not* 1e-6)


# ----------------------------------------------------------------------------- input slice (L3)
def dequantize(q, max_quantized_value=2.0, min_quantized_value=-2.0, dtype=np.float64):
    """W/utils.py:23-38: q*(range/255) + (range/512 + min)."""
    assert max_quantized_value > min_quantized_value
    rng = max_quantized_value - min_quantized_value
    scalar = rng / 255.0
    bias = (rng / 512.0) + min_quantized_value
    return np.asarray(q, dtype=dtype) * dtype(scalar) + dtype(bias)


def get_video_matrix(q_frames, max_frames=300, dtype=np.float64):
    """W/readers.py:159-187: uint8 [n, D] -> dequantised [max_frames, D] (zero padded AFTER
    dequantisation, truncated to max_frames) and num_frames = min(n, max_frames)."""
    q_frames = np.asarray(q_frames)
    n = q_frames.shape[0]
    num_frames = min(n, max_frames)
    x = dequantize(q_frames[:num_frames], dtype=dtype)
    out = np.zeros((max_frames, q_frames.shape[1]), dtype=dtype)
    out[:num_frames] = x
    return out, num_frames


def labels_to_multihot(label_ids, num_classes=4716):
    """W/readers.py:120,217-220: sparse int64 ids -> bool [num_classes]; duplicates and order are
    irrelevant (sparse_to_dense with validate_indices=False)."""
    out = np.zeros((num_classes,), dtype=bool)
    for i in label_ids:
        out[int(i)] = True
    return out


def l2_normalize(x, axis=-1, eps=1e-12):
    """A.9 tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), eps)).  Used by DefaultTransformer
    (W/all_feature_transform/default_transformer.py:4-8) on the last axis."""
    ss = np.sum(x * x, axis=axis, keepdims=True)
    return x / np.sqrt(np.maximum(ss, eps))


def l2_normalize_bwd(x, dy, axis=-1, eps=1e-12):
    """y = x*r, r = rsqrt(max(ss, eps)); dx = r*(dy - y*(y.dy)) if ss>eps
    else r*dy."""
    ss = np.sum(x * x, axis=axis, keepdims=True)
    r = 1.0 / np.sqrt(np.maximum(ss, eps))
    y = x * r
    dot = np.sum(y * dy, axis=axis, keepdims=True)
    return np.where(ss > eps, r * (dy - y * dot), r * dy)


def dequant_l2norm_folded(q, num_frames=None):
    """the dequantise + L2-normalise pair expressed on the raw uint8 rows:
    x_norm = (s*q + b) / sqrt(s^2*sum(q^2) + 2*s*b*sum(q) + D*b^2); rows >= num_frames are 0.
    q: [..., F, D] uint8.  Returns float64.  Used to check the folded-GEMM prologue."""
    q = np.asarray(q)
    s = 4.0 / 255.0
    b = 4.0 / 512.0 - 2.0
    qi = q.astype(np.int64)
    sq = qi.sum(-1, keepdims=True).astype(np.float64)
    sqq = (qi * qi).sum(-1, keepdims=True).astype(np.float64)
    D = q.shape[-1]
    ss = s * s * sqq + 2 * s * b * sq + D * b * b
    x = (s * qi + b) / np.sqrt(np.maximum(ss, 1e-12))
    if num_frames is not None:
        F = q.shape[-2]
        mask = np.arange(F)[None, :] < np.asarray(num_frames)[:, None]
        x = x * mask[..., None]
    return x


# ----------------------------------------------------------------------------- elementwise
def sigmoid(z):
    out = np.empty_like(z)
    pos = z >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-z[pos]))
    ez = np.exp(z[~pos])
    out[~pos] = ez / (1.0 + ez)
    return out


def softmax(z, axis=-1):
    """A.10: max-subtracted softmax."""
    m = np.max(z, axis=axis, keepdims=True)
    e = np.exp(z - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def relu6(x):
    return np.minimum(np.maximum(x, 0.0), 6.0)


# ----------------------------------------------------------------------------- heads (L2)
def fully_connected(x, W, b=None):
    """A.1 slim.fully_connected without activation: x.W (+ b); rank-3 inputs are flattened on
    the leading dims (numpy matmul broadcasting does the same)."""
    y = x @ W
    if b is not None:
        y = y + b
    return y


def logistic_model(x, W, b):
    """W/all_video_models/logistic_model.py:23-25: sigmoid(x.W + b)."""
    return sigmoid(fully_connected(x, W, b))


def moe_model(x, Wg, We, be, num_mixtures):
    """W/all_video_models/moe_model.py:40-64.  Wg: [D, V*(M+1)] (no bias), We: [D, V*M],
    be: [V*M].  Column l*(M+1)+m of Wg is gate m of label l (label-major, mixture-minor);
    gate M is the dummy expert that predicts 0.  Returns p [B, V]."""
    M = num_mixtures
    B = x.shape[0]
    G = fully_connected(x, Wg).reshape(-1, M + 1)
    E = fully_connected(x, We, be).reshape(-1, M)
    g = softmax(G, axis=1)
    e = sigmoid(E)
    p = np.sum(g[:, :M] * e, axis=1)
    return p.reshape(B, -1)


def moe_model_bwd(x, Wg, We, be, num_mixtures, dp, need_dx=True):
    """SURVEY.md Appendix G (autodiff of moe_model.py:54-64):
    dp/dE_m = g_m e_m (1-e_m);  dp/dG_j = g_j (e_j [j<M] - p)."""
    M = num_mixtures
    B = x.shape[0]
    V = Wg.shape[1] // (M + 1)
    G = (x @ Wg).reshape(B, V, M + 1)
    E = (x @ We + be).reshape(B, V, M)
    g = softmax(G, axis=2)
    e = sigmoid(E)
    p = np.sum(g[:, :, :M] * e, axis=2)
    epad = np.concatenate([e, np.zeros((B, V, 1), dtype=e.dtype)], axis=2)
    dG = dp[:, :, None] * g * (epad - p[:, :, None])
    dE = dp[:, :, None] * g[:, :, :M] * e * (1 - e)
    dG = dG.reshape(B, -1)
    dE = dE.reshape(B, -1)
    out = {"dWg": x.T @ dG, "dWe": x.T @ dE, "dbe": dE.sum(0)}
    if need_dx:
        out["dx"] = dG @ Wg.T + dE @ We.T
    return out


def deep_combine_chain_model(x, params, num_layers, num_mixtures, relu_type="relu"):
    """W/all_video_models/deep_combine_chain_model.py:12-85.
    params[name] with the reference variable names ("gates-prediction-%d/weights", ...,
    "relu-%d/weights", "relu-%d/biases", "gates--main/weights", ...).  Returns
    (predictions [B,V], support_predictions [B, L*V])."""
    nxt = x
    supports = []
    for layer in range(num_layers):
        s = "prediction-%d" % layer
        sub = moe_model(nxt, params["gates-%s/weights" % s], params["experts-%s/weights" % s],
                        params["experts-%s/biases" % s], num_mixtures)
        act = fully_connected(sub, params["relu-%d/weights" % layer], params["relu-%d/biases" % layer])
        if relu_type == "elu":
            r = np.where(act > 0, act, np.exp(np.minimum(act, 0)) - 1.0)
        else:
            r = np.maximum(act, 0.0)
        nxt = np.concatenate([nxt, l2_normalize(r, axis=1)], axis=1)
        supports.append(sub)
    main = moe_model(nxt, params["gates--main/weights"], params["experts--main/weights"],
                     params["experts--main/biases"], num_mixtures)
    return main, np.concatenate(supports, axis=1)


# ----------------------------------------------------------------------------- recurrent (L2)
def basic_lstm_step(x_t, c, h, W, b, forget_bias=1.0):
    """A.3 BasicLSTMCell: z = [x_t || h].W + b; i, j, f, o = split(z, 4);
    c' = c*sigmoid(f + forget_bias) + sigmoid(i)*tanh(j); h' = tanh(c')*sigmoid(o)."""
    z = np.concatenate([x_t, h], axis=1) @ W + b
    i, j, f, o = np.split(z, 4, axis=1)
    c_new = c * sigmoid(f + forget_bias) + sigmoid(i) * np.tanh(j)
    h_new = np.tanh(c_new) * sigmoid(o)
    return c_new, h_new


def dynamic_rnn_lstm(x, num_frames, layers, forget_bias=1.0):
    """A.4 MultiRNNCell + A.5 tf.nn.dynamic_rnn (copy-through rule text:
    Z/rnn_residual.py:61-188).  x [B,F,D]; layers = [(W_l, b_l)] with W_l [in_l + H, 4H].
    Zero initial state; for row r at t >= num_frames[r] the emitted output is 0 and the state is
    carried unchanged.  Returns outputs [B,F,H] (top layer) and final (c_l, h_l) per layer."""
    B, F, _ = x.shape
    H = layers[0][1].shape[0] // 4
    dt = x.dtype
    cs = [np.zeros((B, H), dtype=dt) for _ in layers]
    hs = [np.zeros((B, H), dtype=dt) for _ in layers]
    outputs = np.zeros((B, F, H), dtype=dt)
    nf = np.asarray(num_frames)
    for t in range(F):
        live = (t < nf)[:, None]
        inp = x[:, t, :]
        for l, (W, b) in enumerate(layers):
            c_new, h_new = basic_lstm_step(inp, cs[l], hs[l], W, b, forget_bias)
            cs[l] = np.where(live, c_new, cs[l])
            hs[l] = np.where(live, h_new, hs[l])
            inp = h_new
        outputs[:, t, :] = np.where(live, inp, 0.0)
    return outputs, list(zip(cs, hs))


def gru_step(x_t, h, Wg, bg, Wc, bc):
    """tf.contrib.rnn.GRUCell (TF 1.0; not in /root/reference -- call sites W/all_frame_models/gru_pooling_model.py:34-38):
    [r | u] = sigmoid([x_t || h].Wg + bg) (bias initialised to 1), c = tanh([x_t || r*h].Wc + bc), h' = u*h + (1-u)*c."""
    ru = sigmoid(np.concatenate([x_t, h], axis=1) @ Wg + bg)
    r, u = np.split(ru, 2, axis=1)
    c = np.tanh(np.concatenate([x_t, r * h], axis=1) @ Wc + bc)
    return u * h + (1.0 - u) * c


def dynamic_rnn_gru(x, num_frames, layers):
    """MultiRNNCell([GRUCell]) under tf.nn.dynamic_rnn with the copy-through rule of dynamic_rnn_lstm.
    layers = [(Wg [in+H, 2H], bg [2H], Wc [in+H, H], bc [H])].  Returns outputs [B,F,H] (top layer), [h_l final]."""
    B, F, _ = x.shape
    H = layers[0][3].shape[0]
    hs = [np.zeros((B, H), dtype=x.dtype) for _ in layers]
    outputs = np.zeros((B, F, H), dtype=x.dtype)
    nf = np.asarray(num_frames)
    for t in range(F):
        live = (t < nf)[:, None]
        inp = x[:, t, :]
        for l, (Wg, bg, Wc, bc) in enumerate(layers):
            h_new = gru_step(inp, hs[l], Wg, bg, Wc, bc)
            hs[l] = np.where(live, h_new, hs[l])
            inp = h_new
        outputs[:, t, :] = np.where(live, inp, 0.0)
    return outputs, hs


def layer_norm(x, gamma, beta, eps=1e-12):
    """tf.contrib.layers.layer_norm on [B, H] (TF 1.0): moments over the last axis, variance_epsilon 1e-12."""
    mean = x.mean(axis=1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=1, keepdims=True)
    return (x - mean) / np.sqrt(var + eps) * gamma + beta


def layer_norm_lstm_step(x_t, c, h, W, gammas, betas, forget_bias=1.0):
    """tf.contrib.rnn.LayerNormBasicLSTMCell (TF 1.0; call site W/all_frame_models/layernorm_lstm_memory_model.py:37-50), no
    dropout: [i|j|f|o] = [x_t || h].W (no bias), each normalised; c' = LN(c*sigmoid(f + fb) + sigmoid(i)*tanh(j)); h' =
    tanh(c')*sigmoid(o).  gammas / betas: input, transform, forget, output, state."""
    i, j, f, o = np.split(np.concatenate([x_t, h], axis=1) @ W, 4, axis=1)
    i, j, f, o = [layer_norm(v, gammas[k], betas[k]) for k, v in enumerate((i, j, f, o))]
    c_new = layer_norm(c * sigmoid(f + forget_bias) + sigmoid(i) * np.tanh(j), gammas[4], betas[4])
    return c_new, np.tanh(c_new) * sigmoid(o)


def dynamic_rnn_layer_norm_lstm(x, num_frames, layers, forget_bias=1.0):
    """layers = [(W [in+H, 4H], [gamma]*5, [beta]*5)].  Returns outputs [B,F,H] (top layer), [(c_l, h_l) final]."""
    B, F, _ = x.shape
    H = layers[0][1][0].shape[0]
    cs = [np.zeros((B, H), dtype=x.dtype) for _ in layers]
    hs = [np.zeros((B, H), dtype=x.dtype) for _ in layers]
    outputs = np.zeros((B, F, H), dtype=x.dtype)
    nf = np.asarray(num_frames)
    for t in range(F):
        live = (t < nf)[:, None]
        inp = x[:, t, :]
        for l, (W, ga, be) in enumerate(layers):
            c_new, h_new = layer_norm_lstm_step(inp, cs[l], hs[l], W, ga, be, forget_bias)
            cs[l] = np.where(live, c_new, cs[l])
            hs[l] = np.where(live, h_new, hs[l])
            inp = h_new
        outputs[:, t, :] = np.where(live, inp, 0.0)
    return outputs, list(zip(cs, hs))


def lstm_model_state(x, num_frames, layers):
    """W/all_frame_models/lstm_model.py:34-52: state_is_tuple=False => the head input is the
    whole state [c0 || h0 || c1 || h1] (4H for two layers)."""
    _, finals = dynamic_rnn_lstm(x, num_frames, layers)
    return np.concatenate([np.concatenate([c, h], axis=1) for c, h in finals], axis=1)


def lstm_memory_model_state(x, num_frames, layers):
    """W/all_frame_models/lstm_memory_model.py:57-61: concat of the c states only."""
    _, finals = dynamic_rnn_lstm(x, num_frames, layers)
    return np.concatenate([c for c, _ in finals], axis=1)


def attention_pool(x, outputs, num_frames, Wa, ba):
    """W/all_frame_models/lstm_attention_max_pooling_model.py:34,51-63.
    logits = [x || outputs].Wa + ba  [B,F,A]; softmax over FRAMES (all F, incl. padding);
    multiply by the sequence mask; renormalise over frames; pooled[b,a,:] = sum_f w[b,a,f] out[b,f,:]."""
    B, F, _ = x.shape
    mask = (np.arange(F)[None, :] < np.asarray(num_frames)[:, None]).astype(x.dtype)
    act = fully_connected(np.concatenate([x, outputs], axis=2), Wa, ba)      # [B,F,A]
    sm = softmax(act, axis=1)
    w = np.einsum("ijk,ij->ikj", sm, mask)                                   # [B,A,F]
    w = w / np.sum(w, axis=2, keepdims=True)
    pooled = np.einsum("ijk,ilj->ilk", outputs, w)                           # [B,A,H]
    return pooled, w


def lstm_attention_max_pooling_model(x, num_frames, layers, Wa, ba, Wg, We, be, num_mixtures):
    """lstm_attention_max_pooling_model.py:13-66: LSTM -> attention pooling -> MoE per attention
    -> max over attentions."""
    outputs, _ = dynamic_rnn_lstm(x, num_frames, layers)
    pooled, _ = attention_pool(x, outputs, num_frames, Wa, ba)
    B, A, H = pooled.shape
    p = moe_model(pooled.reshape(B * A, H), Wg, We, be, num_mixtures).reshape(B, A, -1)
    return p.max(axis=1)


def sample_random_frames(x, num_frames, u):
    """W/model_utils.py:51-70: frame_index = int(u * num_frames), u ~ U[0,1) of shape [B, S]."""
    nf = np.asarray(num_frames, dtype=np.float32)[:, None]
    idx = (np.asarray(u, dtype=np.float32) * nf).astype(np.int32)
    return np.take_along_axis(x, idx[:, :, None], axis=1), idx


def sample_random_sequence(x, num_frames, u, num_samples):
    """W/model_utils.py:23-48: start = int(u * (max(nf - S, 0) + 1)); idx = min(start + arange(S), nf-1)."""
    nf = np.asarray(num_frames, dtype=np.int32)[:, None]
    max_start = np.maximum(nf - num_samples, 0)
    start = (np.asarray(u, dtype=np.float32).reshape(-1, 1) * (max_start + 1).astype(np.float32)).astype(np.int32)
    idx = np.minimum(start + np.arange(num_samples, dtype=np.int32)[None, :], nf - 1)
    return np.take_along_axis(x, idx[:, :, None], axis=1), idx


def batch_norm_train(x, gamma, beta, eps=1e-3):
    """A.11 slim.batch_norm(is_training=True): batch mean / biased variance on axis 0."""
    mu = x.mean(0)
    var = x.var(0)
    return gamma * (x - mu) / np.sqrt(var + eps) + beta, mu, var


def dbof_model_hidden(x_sampled, Wc, bc, Wh, bh, pooling="max"):
    """W/all_frame_models/dbof_model.py:57-116 with add_batch_norm=False: per-frame cluster FC ->
    relu6 -> pool over frames -> hidden FC -> relu6.  x_sampled [B,S,D] already frame-sampled."""
    B, S, D = x_sampled.shape
    act = relu6(x_sampled.reshape(-1, D) @ Wc + bc).reshape(B, S, -1)
    pooled = act.max(axis=1) if pooling == "max" else act.mean(axis=1)
    return relu6(pooled @ Wh + bh)


def dbof_model_bn(x_sampled, P, pooling="max", eps=1e-3):
    """W/all_frame_models/dbof_model.py:57-116 with add_batch_norm=True (the default, :52) on already sampled frames [B,S,D]:
    input_bn (:66-71) -> cluster matmul (:73-78, no bias under batch norm) -> cluster_bn (:79-84) -> relu6 (:91) -> pooling over
    frames (:95-96) -> hidden matmul (:98-102) -> hidden1_bn (:103-108) -> relu6 (:115).  P as in torch_ref.dbof_model_bn."""
    B, S, D = x_sampled.shape
    r, _, _ = batch_norm_train(x_sampled.reshape(-1, D), P["input_bn/gamma"], P["input_bn/beta"], eps)
    a, _, _ = batch_norm_train(r @ P["Variable"], P["cluster_bn/gamma"], P["cluster_bn/beta"], eps)
    a = relu6(a).reshape(B, S, -1)
    pooled = a.max(axis=1) if pooling == "max" else a.mean(axis=1)
    h, _, _ = batch_norm_train(pooled @ P["Variable_1"], P["hidden1_bn/gamma"], P["hidden1_bn/beta"], eps)
    return relu6(h)


def netvlad(x, num_frames, Wc, bc, centres, eps=1e-12):
    """SURVEY.md Appendix B (NOT in the reference): soft-assignment + residual aggregation +
    intra-normalisation + L2.  x [B,F,D] (already normalised), Wc [D,K], bc [K], centres [K,D].
    Returns v [B, K*D] and the assignment a [B,F,K]."""
    B, F, D = x.shape
    mask = (np.arange(F)[None, :] < np.asarray(num_frames)[:, None]).astype(x.dtype)
    a = softmax(x @ Wc + bc, axis=2) * mask[:, :, None]
    n = a.sum(axis=1)                                                # [B,K]
    vlad = np.einsum("bfk,bfd->bkd", a, x) - n[:, :, None] * centres[None]
    vlad = l2_normalize(vlad, axis=2, eps=eps)
    v = l2_normalize(vlad.reshape(B, -1), axis=1, eps=eps)
    return v, a


def netvlad_hidden(x, num_frames, Wc, bc, centres, Wh, bh, Wgate=None, bgate=None):
    """Appendix B: h = v.Wh + bh; gated: h * sigmoid(h.Wgate + bgate)."""
    v, _ = netvlad(x, num_frames, Wc, bc, centres)
    h = v @ Wh + bh
    if Wgate is not None:
        h = h * sigmoid(h @ Wgate + bgate)
    return h


# ----------------------------------------------------------------------------- loss (L1)
def label_smoothing(labels, epsilon=0.1):
    """W/losses.py:46-54: y*(1-eps) + (sum_l y / K)*eps."""
    y = labels.astype(np.float64) if labels.dtype == bool else labels
    prior = y.sum(axis=1, keepdims=True) / y.shape[1]
    return y * (1.0 - epsilon) + prior * epsilon


def cross_entropy_loss(p, labels, weights=None, eps=XENT_EPS):
    """W/losses.py:114-130: probability-space cross-entropy, sum over classes, mean over batch."""
    y = labels.astype(p.dtype)
    ce = -(y * np.log(p + eps) + (1 - y) * np.log(1 - p + eps))
    if weights is not None:
        ce = ce * np.asarray(weights, dtype=p.dtype)[:, None]
    return ce.sum(axis=1).mean()


def cross_entropy_loss_bwd(p, labels, weights=None, eps=XENT_EPS, upstream=1.0):
    """Appendix G: dL/dp = -(1/B) (y/(p+eps) - (1-y)/(1-p+eps)) [* w_b]."""
    y = labels.astype(p.dtype)
    B = p.shape[0]
    d = -(y / (p + eps) - (1 - y) / (1 - p + eps)) / B
    if weights is not None:
        d = d * np.asarray(weights, dtype=p.dtype)[:, None]
    return d * upstream


def multitask_cross_entropy_loss(p, support_p, labels, support_labels, support_loss_percent=0.1):
    """W/losses.py:271-279."""
    return (cross_entropy_loss(p, labels) * (1.0 - support_loss_percent)
            + cross_entropy_loss(support_p, support_labels) * support_loss_percent)


def load_vertical_mapping(lines, num_classes, num_verticals):
    """W/losses.py:233-243: `lines` of text; a line of exactly two integers "class vertical" sets vm[class, vertical] = 1."""
    vm = np.zeros((num_classes, num_verticals), dtype=np.float64)
    for line in lines:
        group = [int(t) for t in line.strip().split()]
        if len(group) == 2:
            vm[group[0], group[1]] = 1
    return vm


def get_support_label_type(labels, support_type, num_frequents=200, vertical_mapping=None):
    """W/losses.py:221-257 for a comma list of "label" (:251-253), "frequent" (:246-250: the first num_frequents classes) and
    "vertical" (:229-245: labels . vm > 0.2 with the 0/1 class -> vertical table)."""
    outs = []
    for st in support_type.split(","):
        if st == "label":
            outs.append(labels.astype(np.float64))
        elif st == "frequent":
            outs.append(labels[:, :num_frequents].astype(np.float64))
        elif st == "vertical":
            outs.append((labels.astype(np.float64) @ np.asarray(vertical_mapping, dtype=np.float64) > 0.2).astype(np.float64))
        else:
            raise NotImplementedError(st)
    return np.concatenate(outs, axis=1)


# ----------------------------------------------------------------------------- optimiser (L1)
def exponential_decay(base_lr, global_step, batch_size, decay_examples=4000000, decay=0.95):
    """A.7 / W/train.py:303-308 (staircase=True): lr0 * decay^floor(step*B / decay_examples)."""
    return base_lr * decay ** np.floor(global_step * batch_size / float(decay_examples))


def l2_reg_loss(weights, l2_penalty=1e-8):
    """A.2: sum over regularised weights of l2 * 0.5 * sum(W^2)  (W/train.py:440-442)."""
    return sum(l2_penalty * 0.5 * np.sum(w * w) for w in weights)


def clip_by_norm(g, clip=1.0):
    """A.8 tf.clip_by_norm per tensor (W/utils.py:164-174): g * clip / max(||g||, clip)."""
    n = np.sqrt(np.sum(g * g))
    return g * clip / np.maximum(n, clip)


def adam_step(theta, m, v, g, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """A.6 tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); eps OUTSIDE the sqrt."""
    lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v


def train_step_update(params, grads_data, state, step, base_lr, batch_size, regularised,
                      l2_penalty=1e-8, clip=1.0, decay_examples=4000000, decay=0.95):
    """Optimiser slice of build_graph (W/train.py:301-311,435-466): gradient of
    final_loss = data gradient + l2*w for regularised tensors; per-tensor clip; Adam with the
    staircase LR.  `step` is the 0-based global_step before the update; Adam's t = step + 1."""
    lr = exponential_decay(base_lr, step, batch_size, decay_examples, decay)
    t = step + 1
    new_params, new_state = {}, {}
    for k, w in params.items():
        g = grads_data[k] + (l2_penalty * w if k in regularised else 0.0)
        g = clip_by_norm(g, clip) if clip > 0 else g
        m, v = state.get(k, (np.zeros_like(w), np.zeros_like(w)))
        w2, m2, v2 = adam_step(w, m, v, g, lr, t)
        new_params[k] = w2
        new_state[k] = (m2, v2)
    return new_params, new_state
