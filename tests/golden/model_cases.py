"""Seeded tiny cases shared by tests/golden/make_models_golden.py (writes the expected outputs) and the tests that replay
them (tests/test_golden_models.py).  Inputs are regenerated from the seed -- only the expected OUTPUTS live in the fixture."""
import numpy as np


def case_inputs(name):
    rs = np.random.RandomState({"moe": 101, "chain": 102, "lstm": 103, "gru": 104, "lnlstm": 105, "netvlad": 106, "xent": 107,
                                "logistic": 108, "attention": 109, "dbof_bn": 110, "multitask": 111, "trainstep": 112}[name])
    B, F, D, H, V = 4, 7, 6, 8, 11
    nf = np.array([7, 1, 4, 0])
    if name == "moe":
        return dict(x=rs.randn(B, 16), Wg=rs.randn(16, V * 3) * 0.3, We=rs.randn(16, V * 2) * 0.3, be=rs.randn(V * 2) * 0.1, M=2)
    if name == "chain":
        L, C, Din = 2, 5, 16
        P, d = {}, Din
        for i in range(L):
            s = "prediction-%d" % i
            P["gates-%s/weights" % s] = rs.randn(d, V * 3) * 0.3
            P["experts-%s/weights" % s] = rs.randn(d, V * 2) * 0.3
            P["experts-%s/biases" % s] = rs.randn(V * 2) * 0.1
            P["relu-%d/weights" % i] = rs.randn(V, C) * 0.3
            P["relu-%d/biases" % i] = rs.randn(C) * 0.1
            d += C
        P["gates--main/weights"] = rs.randn(d, V * 3) * 0.3
        P["experts--main/weights"] = rs.randn(d, V * 2) * 0.3
        P["experts--main/biases"] = rs.randn(V * 2) * 0.1
        return dict(x=rs.randn(B, Din), P=P, L=L, M=2)
    if name == "logistic":                       # W/all_video_models/logistic_model.py:12-26
        return dict(x=rs.randn(B, 16), W=rs.randn(16, V) * 0.3, b=rs.randn(V) * 0.1)
    if name == "multitask":                      # W/losses.py:222-279 with support_type "label,frequent"
        return dict(p=rs.rand(B, V) * 0.98 + 0.01, sp=rs.rand(B, V + 5) * 0.98 + 0.01, y=(rs.rand(B, V) < 0.25),
                    support_type="label,frequent", num_frequents=5, percent=0.1)
    if name == "trainstep":                      # W/train.py:301-311,435-466: staircase LR past its first decay, clip hit and not hit
        P = {"fully_connected/weights": rs.randn(16, V) * 0.3, "fully_connected/biases": rs.randn(V) * 0.1}
        G = {"fully_connected/weights": rs.randn(16, V) * 0.5, "fully_connected/biases": rs.randn(V) * 0.01}
        st = {k: (rs.randn(*v.shape) * 0.01, rs.rand(*v.shape) * 0.001) for k, v in P.items()}
        return dict(P=P, G=G, state=st, step=4500, base_lr=0.01, batch_size=1024, regularised=["fully_connected/weights"],
                    l2=1e-3, clip=1.0)
    if name == "dbof_bn":                        # already sampled frames: W/all_frame_models/dbof_model.py:57-116, add_batch_norm=True
        S, C, Hd = 5, 9, 6
        P = {"input_bn/gamma": rs.rand(D) + 0.5, "input_bn/beta": rs.randn(D) * 0.1, "Variable": rs.randn(D, C) * 0.6,
             "cluster_bn/gamma": rs.rand(C) + 0.5, "cluster_bn/beta": rs.randn(C) * 0.3 + 0.5, "Variable_1": rs.randn(C, Hd) * 0.6,
             "hidden1_bn/gamma": rs.rand(Hd) + 0.5, "hidden1_bn/beta": rs.randn(Hd) * 0.3 + 0.5}
        return dict(xs=rs.randn(B, S, D), P=P)
    x = rs.randn(B, F, D) * (np.arange(F)[None, :, None] < nf[:, None, None])
    if name == "attention":                      # W/all_frame_models/lstm_attention_max_pooling_model.py:13-66 (no empty video: 0/0 there)
        nf = np.array([7, 1, 4, 3])
        x = rs.randn(B, F, D) * (np.arange(F)[None, :, None] < nf[:, None, None])
        layers, d, A = [], D, 3
        for _ in range(2):
            layers.append((rs.randn(d + H, 4 * H) * 0.4, rs.randn(4 * H) * 0.1))
            d = H
        return dict(x=x, nf=nf, layers=layers, Wa=rs.randn(D + H, A) * 0.5, ba=rs.randn(A) * 0.1, Wg=rs.randn(H, V * 3) * 0.3,
                    We=rs.randn(H, V * 2) * 0.3, be=rs.randn(V * 2) * 0.1, M=2)
    if name == "lstm":
        layers, d = [], D
        for _ in range(2):
            layers.append((rs.randn(d + H, 4 * H) * 0.4, rs.randn(4 * H) * 0.1))
            d = H
        return dict(x=x, nf=nf, layers=layers)
    if name == "gru":
        layers, d = [], D
        for _ in range(2):
            layers.append((rs.randn(d + H, 2 * H) * 0.4, rs.randn(2 * H) * 0.1 + 1, rs.randn(d + H, H) * 0.4, rs.randn(H) * 0.1))
            d = H
        return dict(x=x, nf=nf, layers=layers)
    if name == "lnlstm":
        layers, d = [], D
        for _ in range(2):
            layers.append((rs.randn(d + H, 4 * H) * 0.4, [rs.rand(H) + 0.5 for _ in range(5)], [rs.randn(H) * 0.2 for _ in range(5)]))
            d = H
        return dict(x=x, nf=nf, layers=layers)
    if name == "netvlad":
        K = 4
        xn = x / np.maximum(np.sqrt((x ** 2).sum(-1, keepdims=True)), 1e-6)
        return dict(x=xn, nf=np.array([7, 1, 4, 3]), Wc=rs.randn(D, K) * 0.5, bc=rs.randn(K) * 0.1, centres=rs.randn(K, D) * 0.3)
    if name == "xent":
        return dict(p=rs.rand(B, V) * 0.98 + 0.01, y=(rs.rand(B, V) < 0.2).astype(np.float64), w=rs.rand(B) + 0.5)
    raise KeyError(name)


def case_outputs(name, np_ref):
    c = case_inputs(name)
    if name == "moe":
        return {"predictions": np_ref.moe_model(c["x"], c["Wg"], c["We"], c["be"], c["M"])}
    if name == "chain":
        main, sup = np_ref.deep_combine_chain_model(c["x"], c["P"], c["L"], c["M"])
        return {"predictions": main, "support_predictions": sup}
    if name == "lstm":
        out, fin = np_ref.dynamic_rnn_lstm(c["x"], c["nf"], c["layers"])
        return {"outputs": out, "state": np_ref.lstm_model_state(c["x"], c["nf"], c["layers"])}
    if name == "gru":
        out, hs = np_ref.dynamic_rnn_gru(c["x"], c["nf"], c["layers"])
        return {"outputs": out, "h0": hs[0], "h1": hs[1]}
    if name == "lnlstm":
        out, fin = np_ref.dynamic_rnn_layer_norm_lstm(c["x"], c["nf"], c["layers"])
        return {"outputs": out, "c0": fin[0][0], "c1": fin[1][0], "h1": fin[1][1]}
    if name == "netvlad":
        v, a = np_ref.netvlad(c["x"], c["nf"], c["Wc"], c["bc"], c["centres"])
        return {"vlad": v, "assignment": a}
    if name == "xent":
        return {"loss": np.asarray(np_ref.cross_entropy_loss(c["p"], c["y"])),
                "weighted_loss": np.asarray(np_ref.cross_entropy_loss(c["p"], c["y"], weights=c["w"])),
                "grad": np_ref.cross_entropy_loss_bwd(c["p"], c["y"])}
    if name == "logistic":
        return {"predictions": np_ref.logistic_model(c["x"], c["W"], c["b"])}
    if name == "attention":
        out, _ = np_ref.dynamic_rnn_lstm(c["x"], c["nf"], c["layers"])
        pooled, w = np_ref.attention_pool(c["x"], out, c["nf"], c["Wa"], c["ba"])
        return {"pooled": pooled, "weights": w,
                "predictions": np_ref.lstm_attention_max_pooling_model(c["x"], c["nf"], c["layers"], c["Wa"], c["ba"], c["Wg"], c["We"],
                                                                        c["be"], c["M"])}
    if name == "dbof_bn":
        return {"hidden": np_ref.dbof_model_bn(c["xs"], c["P"])}
    if name == "multitask":
        sl = np_ref.get_support_label_type(c["y"], c["support_type"], num_frequents=c["num_frequents"])
        return {"support_labels": sl,
                "loss": np.asarray(np_ref.multitask_cross_entropy_loss(c["p"], c["sp"], c["y"], sl, c["percent"]))}
    if name == "trainstep":
        newP, newS = np_ref.train_step_update(c["P"], c["G"], c["state"], c["step"], c["base_lr"], c["batch_size"], set(c["regularised"]),
                                              l2_penalty=c["l2"], clip=c["clip"])
        o = {"lr": np.asarray(np_ref.exponential_decay(c["base_lr"], c["step"], c["batch_size"]))}
        for k in sorted(newP):
            o["param:" + k] = newP[k]
            o["m:" + k], o["v:" + k] = newS[k]
        return o
    raise KeyError(name)


CASES = ["moe", "chain", "lstm", "gru", "lnlstm", "netvlad", "xent", "logistic", "attention", "dbof_bn", "multitask", "trainstep"]
