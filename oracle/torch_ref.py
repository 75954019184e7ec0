"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- second, independent restatement in torch-CPU.

Purpose: (1) gradients by autograd (float64) for backward-parity tests of the HIP kernels,
(2) cross-check of oracle/np_ref.py (two restatements must agree), (3) the `cpu_baseline` "port" leg
of bench.py (float32, all host cores).  **Parity unpinned** (see oracle/__init__.py).

Citations: W = /root/reference/youtube-8m-wangheda/, "A.n" = SURVEY.md Appendix A.
"""
import math

import torch

XENT_EPS = 10e-6  # W/losses.py:115


def dequantize(q, dtype=torch.float32):
    """W/utils.py:23-38."""
    return q.to(dtype) * (4.0 / 255.0) + (4.0 / 512.0 - 2.0)


def l2_normalize(x, dim=-1, eps=1e-12):
    """A.9."""
    return x * torch.rsqrt(torch.clamp((x * x).sum(dim, keepdim=True), min=eps))


def moe(x, Wg, We, be, M):
    """W/all_video_models/moe_model.py:40-64."""
    B = x.shape[0]
    V = We.shape[1] // M
    g = torch.softmax((x @ Wg).view(B, V, M + 1), dim=2)
    e = torch.sigmoid((x @ We + be).view(B, V, M))
    return (g[:, :, :M] * e).sum(2)


def moe_fast(x, Wg, We, be, M):
    """Same function as moe() written with per-mixture [B,V] slices instead of a softmax over a size-(M+1) last axis
    (torch-CPU's softmax / sigmoid on tiny inner dims are an order of magnitude slower); used by the cpu_baseline leg."""
    B = x.shape[0]
    V = We.shape[1] // M
    G = (x @ Wg).view(B, V, M + 1)
    E = (x @ We + be).view(B, V, M)
    gs = [G[:, :, m] for m in range(M + 1)]
    mx = gs[0]
    for t in gs[1:]:
        mx = torch.maximum(mx, t)
    ex = [torch.exp(t - mx) for t in gs]
    den = ex[0]
    for t in ex[1:]:
        den = den + t
    num = ex[0] * torch.sigmoid(E[:, :, 0])
    for m in range(1, M):
        num = num + ex[m] * torch.sigmoid(E[:, :, m])
    return num / den


def logistic(x, W, b):
    """W/all_video_models/logistic_model.py:23-25."""
    return torch.sigmoid(x @ W + b)


def dropout(x, keep_prob, seed, offset=0):
    """tf.nn.dropout with the build's Philox convention (oracle/philox.py): element index = row-major position."""
    from . import philox
    m = torch.from_numpy(philox.dropout_mask(x.numel(), keep_prob, seed, offset)).view(x.shape)
    return torch.where(m, x / torch.tensor(keep_prob, dtype=x.dtype), torch.zeros_like(x))


def deep_combine_chain(x, P, L, M, relu_type="relu", dropout_spec=None, bf16_heads=False):
    """W/all_video_models/deep_combine_chain_model.py:12-85.  dropout_spec = (keep_prob, [seed per sub-model]): :57-58.
    bf16_heads: VALUE emulation of --compute_dtype=bfloat16 on the MoE heads (both GEMM operands rounded to bf16, fp32+
    accumulation), the products that take the bf16 MFMA path at >= 512 rows."""
    if bf16_heads:
        moe_ = lambda x_, Wg, We, be, M_: moe(bf16_round(x_), bf16_round(Wg), bf16_round(We), be, M_)
    else:
        moe_ = moe
    cur, sup = x, []
    for i in range(L):
        s = "prediction-%d" % i
        inp = cur if dropout_spec is None else dropout(cur, dropout_spec[0], dropout_spec[1][i])
        sp = moe_(inp, P["gates-%s/weights" % s], P["experts-%s/weights" % s], P["experts-%s/biases" % s], M)
        a = sp @ P["relu-%d/weights" % i] + P["relu-%d/biases" % i]
        r = torch.nn.functional.elu(a) if relu_type == "elu" else torch.relu(a)
        cur = torch.cat([cur, l2_normalize(r, 1)], 1)
        sup.append(sp)
    main = moe_(cur, P["gates--main/weights"], P["experts--main/weights"], P["experts--main/biases"], M)
    return main, torch.cat(sup, 1)


def bf16_round(t):
    """round-to-nearest-even fp32 -> bf16 -> back (value emulation of the device casts), any float dtype in / same dtype out"""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


def lstm_stack(x, num_frames, layers, forget_bias=1.0, dropout_spec=None, bf16_operands=False):
    """A.3-A.5 (BasicLSTMCell / MultiRNNCell / dynamic_rnn with copy-through; Z/rnn_residual.py:61-188).
    dropout_spec = (input_keep_prob, [seed per layer]): DropoutWrapper(cell, input_keep_prob)
    (W/all_frame_models/lstm_memory_model.py:36-45) -- the layer input of step t is element block t of a time-major
    [F,B,in] mask tensor."""
    B, F, _ = x.shape
    H = layers[0][1].numel() // 4
    c = [x.new_zeros(B, H) for _ in layers]
    h = [x.new_zeros(B, H) for _ in layers]
    outs = []
    for t in range(F):
        live = (t < num_frames).unsqueeze(1)
        inp = x[:, t]
        for l, (W, b) in enumerate(layers):
            if dropout_spec is not None:
                inp = dropout(inp, dropout_spec[0], dropout_spec[1][l], offset=t * inp.numel())
            if bf16_operands == "input":   # --compute_dtype=bfloat16, hoisted input projection only (H % 256 != 0)
                d = inp.shape[1]
                z = bf16_round(inp) @ bf16_round(W[:d]) + h[l] @ W[d:] + b
            elif bf16_operands:   # both products take bf16-rounded operands, fp32+ accumulation
                z = bf16_round(torch.cat([inp, h[l]], 1)) @ bf16_round(W) + b
            else:
                z = torch.cat([inp, h[l]], 1) @ W + b
            i, j, f, o = z.chunk(4, 1)
            cn = c[l] * torch.sigmoid(f + forget_bias) + torch.sigmoid(i) * torch.tanh(j)
            hn = torch.tanh(cn) * torch.sigmoid(o)
            c[l] = torch.where(live, cn, c[l])
            h[l] = torch.where(live, hn, h[l])
            inp = hn
        outs.append(torch.where(live, inp, torch.zeros_like(inp)))
    return torch.stack(outs, 1), c, h


def gru_stack(x, num_frames, layers):
    """tf.contrib.rnn.GRUCell (TF 1.0: gates = sigmoid([x|h].Wg + bg) split r | u; c = tanh([x | r*h].Wc + bc);
    h' = u*h + (1-u)*c) stacked by MultiRNNCell under dynamic_rnn (W/all_frame_models/gru_pooling_model.py:34-47).
    layers: [(Wg, bg, Wc, bc)].  Returns (top outputs [B,F,H], [h_l final])."""
    B, F, _ = x.shape
    H = layers[0][3].numel()
    h = [x.new_zeros(B, H) for _ in layers]
    outs = []
    for t in range(F):
        live = (t < num_frames).unsqueeze(1)
        inp = x[:, t]
        for l, (Wg, bg, Wc, bc) in enumerate(layers):
            r, u = torch.sigmoid(torch.cat([inp, h[l]], 1) @ Wg + bg).chunk(2, 1)
            c = torch.tanh(torch.cat([inp, r * h[l]], 1) @ Wc + bc)
            hn = u * h[l] + (1 - u) * c
            h[l] = torch.where(live, hn, h[l])
            inp = hn
        outs.append(torch.where(live, inp, torch.zeros_like(inp)))
    return torch.stack(outs, 1), h


def layer_norm(x, gamma, beta, eps=1e-12):
    """tf.contrib.layers.layer_norm on a 2-D input (TF 1.0): moments over axis 1, batch_normalization with epsilon 1e-12."""
    mean = x.mean(1, keepdim=True)
    var = ((x - mean) ** 2).mean(1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def lnlstm_stack(x, num_frames, layers, forget_bias=1.0, dropout_spec=None):
    """tf.contrib.rnn.LayerNormBasicLSTMCell (TF 1.0: concat = [x|h].W without bias; i, j, f, o each layer-normalised;
    g = tanh(j) [dropout(g, keep_prob)]; c' = LN_state(c*sigmoid(f + forget_bias) + sigmoid(i)*g); h' = tanh(c')*sigmoid(o))
    stacked under dynamic_rnn (W/all_frame_models/layernorm_lstm_memory_model.py:37-58).
    layers: [(W, [gamma]*5, [beta]*5)] in the order input, transform, forget, output, state.
    dropout_spec = (keep_prob, [seed per layer]): the candidate of step t is element block t of a [F,B,H] mask tensor."""
    B, F, _ = x.shape
    H = layers[0][1][0].numel()
    c = [x.new_zeros(B, H) for _ in layers]
    h = [x.new_zeros(B, H) for _ in layers]
    outs = []
    for t in range(F):
        live = (t < num_frames).unsqueeze(1)
        inp = x[:, t]
        for l, (W, ga, be) in enumerate(layers):
            i, j, f, o = (torch.cat([inp, h[l]], 1) @ W).chunk(4, 1)
            i, j, f, o = [layer_norm(v, ga[k], be[k]) for k, v in enumerate((i, j, f, o))]
            gg = torch.tanh(j)
            if dropout_spec is not None:
                gg = dropout(gg, dropout_spec[0], dropout_spec[1][l], offset=t * B * H)
            cn = layer_norm(c[l] * torch.sigmoid(f + forget_bias) + torch.sigmoid(i) * gg, ga[4], be[4])
            hn = torch.tanh(cn) * torch.sigmoid(o)
            c[l] = torch.where(live, cn, c[l])
            h[l] = torch.where(live, hn, h[l])
            inp = hn
        outs.append(torch.where(live, inp, torch.zeros_like(inp)))
    return torch.stack(outs, 1), c, h


def lstm_model_state(x, num_frames, layers, bf16_operands=False):
    """W/all_frame_models/lstm_model.py:34-52: [c0||h0||c1||h1]."""
    _, c, h = lstm_stack(x, num_frames, layers, bf16_operands=bf16_operands)
    return torch.cat([t for pair in zip(c, h) for t in pair], 1)


def attention_pool(x, outputs, num_frames, Wa, ba, order="input_first"):
    """W/all_frame_models/lstm_attention_max_pooling_model.py:34,51-63.  order: which tensor comes first in the FC input."""
    F = x.shape[1]
    mask = (torch.arange(F)[None, :] < num_frames[:, None]).to(x.dtype)
    act = torch.cat([x, outputs] if order == "input_first" else [outputs, x], 2) @ Wa + ba
    w = torch.softmax(act, dim=1) * mask[:, :, None]           # [B,F,A]
    w = w / w.sum(1, keepdim=True)
    return torch.einsum("bfh,bfa->bah", outputs, w)


def lstm_attention_max_pooling(x, num_frames, layers, Wa, ba, Wg, We, be, M):
    outputs, _, _ = lstm_stack(x, num_frames, layers)
    pooled = attention_pool(x, outputs, num_frames, Wa, ba)
    B, A, H = pooled.shape
    return moe(pooled.reshape(B * A, H), Wg, We, be, M).view(B, A, -1).max(1).values


def lstm_parallel_finaloutput(x, num_frames, layer_sets, feature_sizes):
    """W/all_frame_models/lstm_parallel_finaloutput_model.py:34-64: split by feature, l2-normalise each part, one LSTM
    stack per part, concat of every layer's final h (state_is_tuple=True -> x.h)."""
    states, off = [], 0
    for fs, layers in zip(feature_sizes, layer_sets):
        sub = l2_normalize(x[:, :, off:off + fs], 2)
        off += fs
        _, _, h = lstm_stack(sub, num_frames, layers)
        states.extend(h)
    return torch.cat(states, 1)


def lstm_positional_attention_max_pooling(x, num_frames, layers, emb, Wa, ba, Wg, We, be, M):
    """W/all_frame_models/lstm_positional_attention_max_pooling_model.py:30-66,68-85."""
    B, F, D = x.shape
    outputs, _, _ = lstm_stack(x, num_frames, layers)
    mask = (torch.arange(F)[None, :] < num_frames[:, None]).to(x.dtype)
    mean_input = torch.einsum("ijk,ij->ik", x, mask) / num_frames.to(x.dtype)[:, None]
    act = torch.cat([x, emb.expand(B, F, emb.shape[2]), mean_input[:, None, :].expand(B, F, D), outputs], 2) @ Wa + ba
    w = torch.softmax(act, dim=1) * mask[:, :, None]
    w = w / w.sum(1, keepdim=True)
    pooled = torch.einsum("bfh,bfa->bah", outputs, w)
    A, H = pooled.shape[1], pooled.shape[2]
    return moe(pooled.reshape(B * A, H), Wg, We, be, M).view(B, A, -1).max(1).values


def cnn_deep_combine_chain(x, num_frames, P, L, M, relu_cells):
    """W/all_frame_models/cnn_deep_combine_chain_model.py:13-88 (cnn: :13-39)."""
    B, F, D = x.shape
    mask = (torch.arange(F)[None, :] < num_frames[:, None]).to(x.dtype)
    mean_input = torch.einsum("ijk,ij->ik", x, mask) / num_frames.to(x.dtype)[:, None]

    def cnn(scope):
        shifts = [x] + [torch.cat([x.new_zeros(B, i, D), x[:, :F - i]], 1) for i in (1, 2)]
        outs = [torch.cat(shifts[:fs], 2) @ P["%scnn-filter-len%d" % (scope, fs)] for fs in (1, 2, 3)]
        return l2_normalize(torch.cat(outs, 2).max(1).values, 1)

    relu_layers = [l2_normalize(torch.relu(mean_input @ P["mean-relu/weights"] + P["mean-relu/biases"]), 1)]
    nxt, sup = cnn("cnn0"), []
    for layer in range(L):
        s = "prediction-%d" % layer
        sp = moe(nxt, P["gates-%s/weights" % s], P["experts-%s/weights" % s], P["experts-%s/biases" % s], M)
        sup.append(sp)
        relu_layers.append(l2_normalize(torch.relu(sp @ P["relu-%d/weights" % layer] + P["relu-%d/biases" % layer]), 1))
        nxt = torch.cat([mean_input, cnn("cnn%d" % (layer + 1))] + relu_layers, 1)
    main = moe(nxt, P["gates--main/weights"], P["experts--main/weights"], P["experts--main/biases"], M)
    return main, torch.cat(sup, 1)


def reform_distill_labels(y, distill, p):
    """W/train.py:320-327."""
    sy = y.sum(1, keepdim=True)
    sd = distill.sum(1, keepdim=True) + 1e-6
    return torch.clamp(y + distill * (sy / sd * p), 0.0, 1.0)


def weights_by_predictions(y, predictions):
    """W/train.py:250-260."""
    eps = 1e-6
    ce = -(y * torch.log(predictions + eps) + (1 - y) * torch.log(1 - predictions + eps)).sum(1)
    return torch.where(ce > (ce + eps).mean(), torch.full_like(ce, 3.0), torch.full_like(ce, 0.5))


def netvlad(x, num_frames, Wc, bc, centres, eps=1e-12):
    """SURVEY.md Appendix B (not in the reference)."""
    B, F, D = x.shape
    mask = (torch.arange(F)[None, :] < num_frames[:, None]).to(x.dtype)
    a = torch.softmax(x @ Wc + bc, dim=2) * mask[:, :, None]
    vlad = torch.einsum("bfk,bfd->bkd", a, x) - a.sum(1)[:, :, None] * centres[None]
    vlad = l2_normalize(vlad, 2, eps)
    return l2_normalize(vlad.reshape(B, -1), 1, eps)


def netvlad_hidden(x, num_frames, Wc, bc, centres, Wh, bh, Wgate=None, bgate=None):
    h = netvlad(x, num_frames, Wc, bc, centres) @ Wh + bh
    if Wgate is not None:
        h = h * torch.sigmoid(h @ Wgate + bgate)
    return h


def gated_netvlad_attention_chain(x, num_frames, P, L, M, A, bf16_heads=False):
    """BASELINE configs[4] composite as SURVEY.md Appendix B fixes it (not a reference class): gated NetVLAD descriptor,
    attention pooling of the frames themselves (lstm_attention_max_pooling_model.py:34,51-63 with outputs := x and the FC
    input [x || mean_x]), DeepCombineChainModel on [h || att_a], max over the A attentions for predictions and support."""
    B, F, D = x.shape
    h = netvlad_hidden(x, num_frames, P["netvlad/cluster_weights"], P["netvlad/cluster_biases"], P["netvlad/centres"],
                       P["netvlad/hidden/weights"], P["netvlad/hidden/biases"], P["netvlad/gating/weights"],
                       P["netvlad/gating/biases"])
    mean_x = (x.sum(1, keepdim=True) / num_frames.to(x.dtype).clamp(min=1).view(B, 1, 1)).expand(B, F, D)
    att = attention_pool(mean_x, x, num_frames, P["attention-/weights"], P["attention-/biases"], order="outputs_first")
    cin = torch.cat([h[:, None, :].expand(B, A, h.shape[1]), att], 2).reshape(B * A, -1)
    main, sup = deep_combine_chain(cin, P, L, M, bf16_heads=bf16_heads)
    return main.view(B, A, -1).max(1).values, sup.view(B, A, -1).max(1).values


def batch_norm_train(x, gamma, beta, eps=1e-3):
    """slim.batch_norm(center=True, scale=True, is_training=True): batch mean / BIASED variance over axis 0.
    Returns (y, mean, var); the moving averages follow mm <- decay mm + (1 - decay) mean (decay 0.999), same for var."""
    mu = x.mean(0)
    var = ((x - mu) ** 2).mean(0)
    return gamma * (x - mu) * torch.rsqrt(var + eps) + beta, mu, var


def dbof_model_bn(xs, P, pooling="max", eps=1e-3):
    """W/all_frame_models/dbof_model.py:57-116 with add_batch_norm=True on already sampled frames xs [B,S,D]: input_bn ->
    cluster FC -> cluster_bn -> relu6 -> pool over frames (amax: the gradient is split between tied maxima, as tf.reduce_max's)
    -> hidden FC -> hidden1_bn -> relu6.  P: "Variable" (cluster weights), "Variable_1" (hidden weights), "<bn>/gamma|beta"."""
    B, S, D = xs.shape
    r, _, _ = batch_norm_train(xs.reshape(-1, D), P["input_bn/gamma"], P["input_bn/beta"], eps)
    a, _, _ = batch_norm_train(r @ P["Variable"], P["cluster_bn/gamma"], P["cluster_bn/beta"], eps)
    a = torch.clamp(a, 0, 6).view(B, S, -1)
    pooled = a.amax(1) if pooling == "max" else a.mean(1)
    h, _, _ = batch_norm_train(pooled @ P["Variable_1"], P["hidden1_bn/gamma"], P["hidden1_bn/beta"], eps)
    return torch.clamp(h, 0, 6)


def dbof_hidden(xs, Wc, bc, Wh, bh, pooling="max"):
    """W/all_frame_models/dbof_model.py:57-116, add_batch_norm=False."""
    B, S, D = xs.shape
    act = torch.clamp(xs.reshape(-1, D) @ Wc + bc, 0, 6).view(B, S, -1)
    pooled = act.amax(1) if pooling == "max" else act.mean(1)      # amax: tied maxima share the gradient (tf.reduce_max)
    return torch.clamp(pooled @ Wh + bh, 0, 6)


def cross_entropy(p, y, weights=None, eps=XENT_EPS):
    """W/losses.py:114-130."""
    y = y.to(p.dtype)
    ce = -(y * torch.log(p + eps) + (1 - y) * torch.log(1 - p + eps))
    if weights is not None:
        ce = ce * weights[:, None]
    return ce.sum(1).mean()


def exponential_decay(base_lr, step, batch, decay_examples=4000000, decay=0.95):
    """A.7 / W/train.py:303-308."""
    return base_lr * decay ** math.floor(step * batch / float(decay_examples))


class TFAdam:
    """A.6 + W/train.py:435-466 + W/utils.py:164-174: grad of (label_loss + sum l2*0.5*|W|^2), per-tensor
    clip_by_norm, TF-1 Adam (eps outside sqrt, bias correction folded into lr_t)."""

    def __init__(self, params, regularised, base_lr=0.01, batch_size=1024, l2=1e-8, clip=1.0,
                 decay_examples=4000000, decay=0.95, b1=0.9, b2=0.999, eps=1e-8):
        self.params = params                  # dict name -> tensor (requires_grad leaf)
        self.reg = set(regularised)
        self.base_lr, self.batch, self.l2, self.clip = base_lr, batch_size, l2, clip
        self.de, self.decay, self.b1, self.b2, self.eps = decay_examples, decay, b1, b2, eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.step_no = 0

    @torch.no_grad()
    def step(self):
        lr = exponential_decay(self.base_lr, self.step_no, self.batch, self.de, self.decay)
        t = self.step_no + 1
        lr_t = lr * math.sqrt(1 - self.b2 ** t) / (1 - self.b1 ** t)
        for k, w in self.params.items():
            g = w.grad
            if k in self.reg:
                g = g + self.l2 * w
            if self.clip > 0:
                g = g * (self.clip / torch.clamp(g.norm(), min=self.clip))
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            w.sub_(lr_t * self.m[k] / (self.v[k].sqrt() + self.eps))
            w.grad = None
        self.step_no += 1


def xavier_uniform_(t, gen):
    """A.1: slim default weights initialiser, uniform +-sqrt(6/(fan_in+fan_out))."""
    lim = math.sqrt(6.0 / (t.shape[0] + t.shape[1]))
    return t.uniform_(-lim, lim, generator=gen)


def make_moe_params(D, V, M, dtype=torch.float32, seed=0):
    gen = torch.Generator().manual_seed(seed)
    P = {"gates/weights": xavier_uniform_(torch.empty(D, V * (M + 1), dtype=dtype), gen),
         "experts/weights": xavier_uniform_(torch.empty(D, V * M, dtype=dtype), gen),
         "experts/biases": torch.zeros(V * M, dtype=dtype)}
    return P


class MoeTrainStepCPU:
    """Whole training step of BASELINE config[1] (MoeModel M=2 on video-level features) on the
    host: transform (W/train.py:343-344) -> MoeModel -> CrossEntropyLoss -> reg -> clip -> Adam."""

    def __init__(self, D=1152, V=4716, M=2, batch_size=1024, dtype=torch.float32, seed=0, base_lr=0.01):
        self.M = M
        self.P = {k: v.requires_grad_(True) for k, v in make_moe_params(D, V, M, dtype, seed).items()}
        self.opt = TFAdam(self.P, ["gates/weights", "experts/weights"], base_lr=base_lr, batch_size=batch_size)

    def step(self, x_raw, labels):
        x = l2_normalize(x_raw, 1)
        p = moe_fast(x, self.P["gates/weights"], self.P["experts/weights"], self.P["experts/biases"], self.M)
        loss = cross_entropy(p, labels)
        loss.backward()
        self.opt.step()
        return loss.detach(), p.detach()


class LstmTrainStepCPU:
    """Whole training step of BASELINE configs[3] on the host (bench.py cpu_baseline leg, kind "port"): dequantise +
    l2-normalise (W/readers.py:178-187, W/train.py:343-344) -> LstmModel (W/all_frame_models/lstm_model.py:15-57: 2 x
    BasicLSTMCell under dynamic_rnn, head on [c0||h0||c1||h1]) -> MoeModel head -> CrossEntropyLoss -> reg -> clip -> Adam."""

    def __init__(self, D=1152, H=1024, L=2, V=4716, M=2, batch_size=128, dtype=torch.float32, seed=0, base_lr=0.01):
        gen = torch.Generator().manual_seed(seed)
        self.M, self.L = M, L
        P = {}
        d_in = D
        for l in range(L):
            P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l] = xavier_uniform_(torch.empty(d_in + H, 4 * H, dtype=dtype), gen)
            P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l] = torch.zeros(4 * H, dtype=dtype)
            d_in = H
        S = 2 * L * H
        P["gates/weights"] = xavier_uniform_(torch.empty(S, V * (M + 1), dtype=dtype), gen)
        P["experts/weights"] = xavier_uniform_(torch.empty(S, V * M, dtype=dtype), gen)
        P["experts/biases"] = torch.zeros(V * M, dtype=dtype)
        self.P = {k: v.requires_grad_(True) for k, v in P.items()}
        self.opt = TFAdam(self.P, ["gates/weights", "experts/weights"], base_lr=base_lr, batch_size=batch_size)
        self.dtype = dtype

    def step(self, q_frames, num_frames, labels):
        x = l2_normalize(dequantize(q_frames, self.dtype), 2)
        layers = [(self.P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l],
                   self.P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l]) for l in range(self.L)]
        state = lstm_model_state(x, num_frames, layers)
        p = moe_fast(state, self.P["gates/weights"], self.P["experts/weights"], self.P["experts/biases"], self.M)
        loss = cross_entropy(p, labels)
        loss.backward()
        self.opt.step()
        return loss.detach(), p.detach()


class LstmTrainStepOneDNN:
    """The same training step as LstmTrainStepCPU with the recurrent stack on torch.nn.LSTM (PyTorch's fused CPU LSTM: one library
    call per layer and direction instead of 300 Python-level cell steps).  tests/test_oracle_thirdparty.py proves nn.LSTM computes
    the restatement's function under the gate re-ordering (i, j, f, o) -> (i, f, g, o) with forget_bias folded into the bias; this
    class only TIMES that form next to the per-frame port (bench.py cpu_baseline: a CPU baseline should not be slow because its
    loop is written in Python).  Parameters live in nn.LSTM's own layout (weight_ih / weight_hh per layer); every video is F
    frames long here (the bench's throughput batches), so no packing."""

    def __init__(self, D=1152, H=1024, L=2, V=4716, M=2, batch_size=32, dtype=torch.float32, seed=0, base_lr=0.01):
        torch.manual_seed(seed)
        self.M = M
        self.lstm = torch.nn.LSTM(D, H, num_layers=L, batch_first=True).to(dtype)
        with torch.no_grad():
            for l in range(L):
                getattr(self.lstm, "bias_ih_l%d" % l)[H:2 * H] += 1.0          # forget_bias = 1 (BasicLSTMCell)
        gen = torch.Generator().manual_seed(seed)
        S = 2 * L * H
        self.P = {"gates/weights": xavier_uniform_(torch.empty(S, V * (M + 1), dtype=dtype), gen).requires_grad_(True),
                  "experts/weights": xavier_uniform_(torch.empty(S, V * M, dtype=dtype), gen).requires_grad_(True),
                  "experts/biases": torch.zeros(V * M, dtype=dtype).requires_grad_(True)}
        allp = dict(self.P)
        for n, p_ in self.lstm.named_parameters():
            allp["lstm/" + n] = p_
        reg = ["gates/weights", "experts/weights"] + ["lstm/" + n for n, _ in self.lstm.named_parameters() if n.startswith("weight")]
        self.opt = TFAdam(allp, reg, base_lr=base_lr, batch_size=batch_size)
        self.dtype = dtype

    def step(self, q_frames, num_frames, labels):
        x = l2_normalize(dequantize(q_frames, self.dtype), 2)
        _, (hn, cn) = self.lstm(x)
        state = torch.cat([t for l in range(hn.shape[0]) for t in (cn[l], hn[l])], 1)       # [c0 || h0 || c1 || h1]
        p = moe_fast(state, self.P["gates/weights"], self.P["experts/weights"], self.P["experts/biases"], self.M)
        loss = cross_entropy(p, labels)
        loss.backward()
        self.opt.step()
        return loss.detach(), p.detach()
