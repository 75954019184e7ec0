"""Variable store: the role TF's graph + variable scopes play in the reference.

The reference builds a graph once; variables are created on first use under scope names
("gates"+sub_scope, "experts"+sub_scope, "RNN/..."; SURVEY.md section 8b "Ownership").  Here
``create_model`` IS the forward pass and runs every step, so ``get_variable`` has get-or-create
semantics keyed by the TF variable name.

MI355X-first layout: after the first forward pass ``Graph.finalize()`` packs every parameter into ONE
flat fp32 arena in HBM, with a parallel gradient arena and Adam m / v arenas.  That makes
  * the gradient all-reduce a handful of large contiguous RCCL calls,
  * clip + Adam two multi-tensor kernel launches over a chunk table (csrc/optim.hip),
  * backward GEMMs write dW straight into their arena slice (beta = 0/1), no memset / accumulate pass.
"""
import contextlib
import math

import torch

ALIGN = 64        # floats: every tensor starts on a 256-byte boundary
CHUNK = 4096      # floats per optimiser chunk (csrc/optim.hip contract)


class Variable(object):
    def __init__(self, name, data, l2=0.0, trainable=True):
        self.name = name
        self.data = data              # torch tensor (view into the arena after finalize)
        self.grad = None              # view into the gradient arena after finalize
        self.l2 = float(l2)           # slim.l2_regularizer coefficient (0 = not regularised)
        self.trainable = trainable
        self.grad_written = False     # first backward write uses beta=0, later ones accumulate
        self.index = -1               # tensor id in the arena tables
        self.offset = -1
        self._graph = None

    @property
    def shape(self):
        return tuple(self.data.shape)

    def numel(self):
        return self.data.numel()

    def grad_beta(self):
        """beta for the next gradient write (0 = overwrite) and marks the slot as written."""
        beta = 1.0 if self.grad_written else 0.0
        if self.grad_written and self._graph is not None and getattr(self._graph, "grad_ready_hook", None) is not None \
                and getattr(self, "_done_reported", False):
            raise RuntimeError("second gradient contribution to %s after grad_done(): under data parallelism its all-reduce "
                               "is already in flight" % self.name)
        act = getattr(self._graph, "early_active", None) if self._graph is not None else None
        done = getattr(self._graph, "early_done", None) if self._graph is not None else None
        if self.grad_written and self.index >= 0 and any(lo <= self.index < hi for lo, hi in (act or []) + (done or [])):
            #  the recurrent stack's early clip + Adam pass has taken this variable's gradient as final: a contribution
            # that arrives now would land in a slot Adam has already consumed and be dropped at the end of the step
            raise RuntimeError("second gradient contribution to %s after its early optimiser update was enqueued" % self.name)
        self.grad_written = True
        return beta

    def grad_done(self):
        """Called by an op when its gradient contribution to this variable is enqueued.  Once every use of this step has reported
        (uses = get_variable calls since begin_step) the graph's hook fires: the data-parallel reducer starts this slice's
        all-reduce, the single-device step starts its clip + Adam on a side stream."""
        self._done_count = getattr(self, "_done_count", 0) + 1
        if self._done_count < getattr(self, "_uses", 1):
            return
        self._done_reported = True
        if self._graph is not None and self._graph.grad_ready_hook is not None:
            self._graph.grad_ready_hook(self)


# ---- initialisers (TF-1.0 defaults) -------------------------------------------------
def xavier_uniform(shape, gen, device):
    fan_in, fan_out = shape[0], shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return torch.empty(shape, dtype=torch.float32, device=device).uniform_(-lim, lim, generator=gen)


def zeros(shape, gen, device):
    return torch.zeros(shape, dtype=torch.float32, device=device)


def ones(shape, gen, device):
    return torch.ones(shape, dtype=torch.float32, device=device)


def random_normal(stddev):
    def init(shape, gen, device):
        return torch.empty(shape, dtype=torch.float32, device=device).normal_(0.0, stddev, generator=gen)
    return init


def random_seed(graph_seed, rank, step, call):
    """Key of random op number `call` of forward pass `step` on `rank`: splitmix64 finaliser over the packed tuple, so that
    tests (and a restarted job) can reproduce every mask from four integers."""
    m = (1 << 64) - 1
    z = (int(graph_seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xBF58476D1CE4E5B9 + int(step) * 0x94D049BB133111EB + int(call) + 1) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


class Graph(object):
    def __init__(self, device=None, seed=0):
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.vars = {}                 # name -> Variable, insertion ordered
        self._scope = []
        self._anon_counter = 0
        self.finalized = False
        self.seed = seed
        self._gen = None
        self.grad_ready_hook = None
        self.token = None              # dummy requires-grad tensor threaded through ops (see ops.py)
        self.rank = 0                  # data-parallel rank: part of every random-op seed (parallel.GradReducer.attach)
        self._rng_step = -1            # forward passes begun so far - 1
        self._rng_calls = 0            # random ops issued in the current forward pass
        # arenas
        self.params = self.grads = self.adam_m = self.adam_v = None
        self.chunks = self.l2 = self.norms = self.partial = None
        self.total = 0
        self.nchunks = 0
        self.wimg = None               # wimg.WeightImages after finalize(): operand images of the weight matrices, kept by Adam
        self.defer_head_dw = False     # set by the recurrent stack's forward: ops before it in the backward pass may defer their dW
        self.side_pending = []         # side streams holding such work in the current backward pass (ops.join_side_work)

    def __del__(self):
        w = getattr(self, "wimg", None)
        if w is not None:
            w.close()                  # the library's lookup table must not outlive the arena it points into

    # -- scoping --------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def _full(self, name):
        return "/".join(self._scope + [name]) if self._scope else name

    def begin_step(self):
        """Resets per-forward state: anonymous tf.Variable() numbering and the grad-written flags."""
        self._anon_counter = 0
        self._rng_step += 1
        self._rng_calls = 0
        self.defer_head_dw = False
        for v in self.vars.values():
            v.grad_written = False
            v._done_reported = False
            v._done_count = 0
            v._uses = 0
        if self.token is None:
            self.token = torch.zeros((), dtype=torch.float32, device=self.device, requires_grad=True)
        if self.wimg is not None:
            self.wimg.begin_step()     # new demands -> resident images; torch-side write to the arena -> refresh

    def next_random_seed(self):
        """64-bit Philox key of the next random op (dropout / noise) of this forward pass."""
        s = random_seed(self.seed, self.rank, self._rng_step, self._rng_calls)
        self._rng_calls += 1
        return s

    def _generator(self):
        if self._gen is None:
            self._gen = torch.Generator(device=self.device)
            self._gen.manual_seed(self.seed)
        return self._gen

    # -- creation / lookup ----------------------------------------------------------------------------
    def get_variable(self, name, shape, initializer=xavier_uniform, l2=0.0, trainable=True):
        full = self._full(name)
        v = self.vars.get(full)
        if v is not None:
            if tuple(v.data.shape) != tuple(shape):
                raise ValueError("variable %s exists with shape %s, requested %s" % (full, tuple(v.data.shape), tuple(shape)))
            v._uses = getattr(v, "_uses", 0) + 1
            return v
        if self.finalized:
            raise RuntimeError("variable %s requested after Graph.finalize(); the model must create the same "
                               "variables on every call" % full)
        data = initializer(tuple(shape), self._generator(), self.device)
        v = Variable(full, data, l2=l2, trainable=trainable)
        v._graph = self
        v._uses = 1
        self.vars[full] = v
        return v

    def anonymous_variable(self, shape, initializer):
        """tf.Variable(...) without a name (W/all_frame_models/dbof_model.py:73-110): "Variable", "Variable_1", ..."""
        n = self._anon_counter
        self._anon_counter += 1
        return self.get_variable("Variable" if n == 0 else "Variable_%d" % n, shape, initializer)

    def trainable_variables(self):
        return [v for v in self.vars.values() if v.trainable]

    # -- arenas -----------------------------------------------------------------------------------------
    def finalize(self):
        """Packs the trainable variables into flat arenas and builds the optimiser chunk table."""
        if self.finalized:
            return
        tv = self.trainable_variables()
        off = 0
        for i, v in enumerate(tv):
            v.index, v.offset = i, off
            off += (v.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        dev = self.device
        self.params = torch.zeros(max(off, ALIGN), dtype=torch.float32, device=dev)
        self.grads = torch.zeros_like(self.params)
        self.adam_m = torch.zeros_like(self.params)
        self.adam_v = torch.zeros_like(self.params)
        chunks, l2 = [], []
        self.chunk_start = []          # first chunk of tensor i; chunk_start[ntensors] = nchunks
        for v in tv:
            self.chunk_start.append(len(chunks))
            n = v.numel()
            dst = self.params[v.offset:v.offset + n].view(v.data.shape)
            dst.copy_(v.data)
            v.data = dst
            v.grad = self.grads[v.offset:v.offset + n].view(v.data.shape)
            l2.append(v.l2)
            for c0 in range(0, n, CHUNK):
                chunks.append((v.offset + c0, min(CHUNK, n - c0), v.index, 0))
        self.nchunks = len(chunks)
        self.chunk_start.append(len(chunks))
        self.chunk_start_dev = torch.tensor(self.chunk_start, dtype=torch.int32).to(dev)
        self.chunks = torch.tensor(chunks if chunks else [(0, 0, 0, 0)], dtype=torch.int32).to(dev).contiguous()
        self.l2 = torch.tensor(l2 if l2 else [0.0], dtype=torch.float32).to(dev)
        self.norms = torch.zeros(max(len(tv), 1), dtype=torch.float32, device=dev)
        self.partial = torch.zeros(max(self.nchunks, 1), dtype=torch.float32, device=dev)
        self.finalized = True
        if self.params.is_cuda:
            from .wimg import WeightImages
            self.wimg = WeightImages(self)

    # -- checkpoint-style access (TF variable names -> arrays) -----------------------------------------
    def state_dict(self):
        return {k: v.data.detach().clone() for k, v in self.vars.items()}

    def load_state_dict(self, sd, strict=True):
        for k, t in sd.items():
            if k not in self.vars:
                if strict:
                    raise KeyError(k)
                continue
            self.vars[k].data.copy_(torch.as_tensor(t, dtype=torch.float32).to(self.device).view(self.vars[k].data.shape))


_default = None


def get_default_graph():
    global _default
    if _default is None:
        _default = Graph()
    return _default


def reset_default_graph(device=None, seed=0):
    global _default
    _default = Graph(device=device, seed=seed)
    return _default


def set_default_graph(g):
    global _default
    _default = g
    return g
