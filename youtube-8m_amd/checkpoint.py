"""Checkpoint interop: safetensors files keyed by the TF variable names
("gates/weights", "experts/biases", "RNN/multi_rnn_cell/cell_0/basic_lstm_cell/weights", ...) with the reference's
shapes, plus Adam slots ("<name>/Adam", "<name>/Adam_1" as tf.train.AdamOptimizer names them) and "global_step".
Rotation like tf.train.Saver(max_to_keep=3) (W/train.py:728).  TF's own Saver-V2 format cannot be read without TF."""
import glob
import os
import re

import torch


def save(train_graph, directory, max_to_keep=3):
    from safetensors.torch import save_file
    from . import seq_ops
    seq_ops.check_persist_errors()              # never checkpoint weights produced by a timed-out recurrence launch
    g = train_graph.graph
    os.makedirs(directory, exist_ok=True)
    sd = {}
    for name, v in g.vars.items():
        sd[name] = v.data.detach().contiguous().cpu()
        if v.trainable and g.finalized:
            n = v.numel()
            sd[name + "/Adam"] = g.adam_m[v.offset:v.offset + n].view(v.shape).detach().cpu().contiguous()
            sd[name + "/Adam_1"] = g.adam_v[v.offset:v.offset + n].view(v.shape).detach().cpu().contiguous()
    sd["global_step"] = torch.tensor([train_graph.global_step], dtype=torch.int64)
    path = os.path.join(directory, "model.ckpt-%d.safetensors" % train_graph.global_step)
    save_file(sd, path)
    ckpts = sorted(glob.glob(os.path.join(directory, "model.ckpt-*.safetensors")),
                   key=lambda p: int(re.search(r"ckpt-(\d+)", p).group(1)))
    for old in ckpts[:-max_to_keep] if max_to_keep else []:
        os.remove(old)
    return path


def latest_checkpoint(directory):
    ckpts = glob.glob(os.path.join(directory, "model.ckpt-*.safetensors"))
    return max(ckpts, key=lambda p: int(re.search(r"ckpt-(\d+)", p).group(1))) if ckpts else None


def restore(train_graph, path):
    """Loads variables (+ Adam slots and global_step when present).  The graph must already hold the variables (one
    forward pass) -- like the reference, which restores into an existing graph."""
    from safetensors.torch import load_file
    g = train_graph.graph
    sd = load_file(path)
    has_slots = any(k.endswith("/Adam") for k in sd)
    if has_slots:
        if not g.vars:
            raise RuntimeError("restore() needs the variables to exist: run one forward pass (TrainGraph.forward) first")
        # the Adam slots live in the arenas: finalize the way step() would (l2 scaling, reducer attach) BEFORE loading them;
        # restoring between the first forward pass and the first step used to drop m / v silently
        if hasattr(train_graph, "ensure_finalized"):
            train_graph.ensure_finalized()
        elif not g.finalized:
            g.finalize()
    for name, v in g.vars.items():
        if name not in sd:
            raise KeyError("checkpoint %s has no variable %s" % (path, name))
        if tuple(sd[name].shape) != tuple(v.shape):
            raise ValueError("shape mismatch for %s: %s vs %s" % (name, tuple(sd[name].shape), tuple(v.shape)))
        v.data.copy_(sd[name].to(v.data.device))
        if v.trainable and has_slots:
            if name + "/Adam" not in sd or name + "/Adam_1" not in sd:
                raise KeyError("checkpoint %s has Adam slots but none for %s" % (path, name))
            n = v.numel()
            g.adam_m[v.offset:v.offset + n].copy_(sd[name + "/Adam"].reshape(-1).to(g.adam_m.device))
            g.adam_v[v.offset:v.offset + n].copy_(sd[name + "/Adam_1"].reshape(-1).to(g.adam_v.device))
    if "global_step" in sd:
        train_graph.global_step = int(sd["global_step"][0])
    return train_graph
