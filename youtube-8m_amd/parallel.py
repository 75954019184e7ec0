"""Synchronous data parallelism: one process per GPU, batch sharded on dim 0, parameters replicated, gradients
summed with RCCL (torch.distributed backend "nccl" on ROCm) over xGMI; the 1/world mean is folded into the
fused clip+Adam pass (gscale), so the reduced arena is consumed in place.

The reference has no synchronous mode (async parameter server over gRPC, W/train.py:624-639,731-776;
SURVEY.md 2.3); SURVEY.md 8e defines this replacement: the N-rank step on shards of a global batch equals the
1-rank step on the whole batch (mean-of-means with equal shards), and the LR staircase uses the GLOBAL batch.

Overlap: ops call Variable.grad_done() as soon as a parameter's gradient slice is final; contiguous finished
slices are all-reduced asynchronously (bucketed, >= bucket_bytes) while backward continues.  xGMI is point-to-point
(7 links x ~153 GB/s per GPU), so few large messages beat many small ones: default bucket 32 MiB.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract); returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(n_global, rank, world):
    """Rows [lo, hi) of the global batch owned by `rank` (equal shards)."""
    if n_global % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (n_global, world))
    per = n_global // world
    return rank * per, (rank + 1) * per


class CabiComm(object):
    """RCCL communicator owned by the C-ABI library (yt8m_comm_*, csrc/comm.hip) -- the form a non-PyTorch host binds
    (INTEGRATION.md).  The 128-byte unique id of rank 0 travels through `exchange`: a callable rank-0-bytes -> bytes
    (e.g. a torch.distributed.TCPStore round trip, a file, MPI); world == 1 needs none."""

    def __init__(self, rank, world, exchange=None, device=None):
        import ctypes
        from . import _lib
        self._lib = _lib
        L = _lib.lib()
        if device is not None:
            torch.cuda.set_device(device)
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(L.yt8m_comm_unique_id(buf))
        if world > 1:
            if exchange is None:
                raise ValueError("world > 1 needs an `exchange` callable for the unique id")
            buf = ctypes.create_string_buffer(exchange(buf.raw if rank == 0 else None), 128)
        self.handle = ctypes.c_void_p()
        _lib.check(L.yt8m_comm_init(int(rank), int(world), buf, ctypes.byref(self.handle)))
        self.rank, self.world = int(rank), int(world)
        self.stream = torch.cuda.Stream()                     # collectives run beside the backward kernels

    def _p(self, t):
        import ctypes
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        return ctypes.c_void_p(t.data_ptr())

    def all_reduce(self, t, mean=False, algo="allreduce"):
        """In place SUM (or mean) of a contiguous fp32 device tensor; returns a handle whose wait() orders the CURRENT stream
        behind the collective.  algo "rs_ag": the same reduction as reduce-scatter + all-gather (yt8m_comm_allreduce_rsag_f32)."""
        import ctypes
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self.stream.wait_event(ready)
        L = self._lib.lib()
        if algo == "rs_ag":
            self._lib.check(L.yt8m_comm_allreduce_rsag_f32(self.handle, self._p(t), t.numel(), int(bool(mean)), 0,
                                                           ctypes.c_void_p(self.stream.cuda_stream)))
        else:
            self._lib.check(L.yt8m_comm_allreduce_f32(self.handle, self._p(t), t.numel(), int(bool(mean)),
                                                      ctypes.c_void_p(self.stream.cuda_stream)))
        done = torch.cuda.Event()
        done.record(self.stream)
        t.record_stream(self.stream)
        return _CabiHandle(done)

    def broadcast(self, t, root=0):
        import ctypes
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self.stream.wait_event(ready)
        self._lib.check(self._lib.lib().yt8m_comm_broadcast_f32(self.handle, self._p(t), t.numel(), int(root),
                                                                ctypes.c_void_p(self.stream.cuda_stream)))
        done = torch.cuda.Event()
        done.record(self.stream)
        torch.cuda.current_stream().wait_event(done)

    def size(self):
        import ctypes
        r, w = ctypes.c_int(-1), ctypes.c_int(-1)
        self._lib.check(self._lib.lib().yt8m_comm_size(self.handle, ctypes.byref(r), ctypes.byref(w)))
        return r.value, w.value

    def close(self):
        if self.handle:
            torch.cuda.synchronize()
            self._lib.check(self._lib.lib().yt8m_comm_destroy(self.handle))
            self.handle = None


class _CabiHandle(object):
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class _Chain(object):
    """Handles of collectives that were enqueued in order: waiting for the last one covers them all."""

    def __init__(self, hs):
        self.hs = hs

    def wait(self):
        for h in self.hs:
            h.wait()


def _rsag_torch(t, group):
    """SUM all-reduce of a contiguous fp32 tensor as reduce-scatter + all-gather through torch.distributed (RCCL); the n % world tail
    takes a small all-reduce.  Backends without reduce-scatter (gloo) fall back to one all-reduce."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) != "nccl":
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
    rank = dist.get_rank(group)
    per = t.numel() // world
    hs = []
    if per > 0:
        body = t[:per * world]
        mine = body[rank * per:(rank + 1) * per]
        hs.append(dist.reduce_scatter_tensor(mine, body, op=dist.ReduceOp.SUM, group=group, async_op=True))
        hs.append(dist.all_gather_into_tensor(body, mine, group=group, async_op=True))
    if t.numel() > per * world:
        hs.append(dist.all_reduce(t[per * world:], op=dist.ReduceOp.SUM, group=group, async_op=True))
    return _Chain(hs)


DP_ALGO = os.environ.get("YT8M_DP_ALGO", "allreduce")              # "allreduce" | "rs_ag"
# CU headroom of the persistent recurrences for RCCL's kernels (0: none).  Measured on one MI355X with a 1-rank RCCL group
# (profiles/r3_force_reducer.md): 32 reserved CUs chain the half-chip backward recurrences one at a time, +2.0 ms on a 24.4 ms step;
# without a reserve a recurrence launched while a collective's kernels hold CUs spins on the part of its grid that is resident
# until they leave -- bounded, never a deadlock (tests/test_gpu_round3.py::test_persistent_recurrences_next_to_a_cu_hogging_kernel).
# The head's all-reduce (386 MB, launched ~0.3 ms into the backward pass) mostly runs beside the first, lone, half-chip recurrence,
# so the default keeps the side-by-side schedule; set 32 on a node where the collectives are slow enough to collide with it.
DP_RESERVED_CUS = int(os.environ.get("YT8M_DP_RESERVED_CUS", "0"))


class GradReducer(object):
    """All-reduces the gradient arena of a variables.Graph; SUM over ranks, mean applied later via gscale.  Transport:
    torch.distributed (backend "nccl" = RCCL) by default, or a CabiComm (the library's own RCCL binding) when `comm` is given.
    algo: "allreduce" (one collective per bucket) or "rs_ag" (reduce-scatter + all-gather per bucket; same result).
    While a reducer is attached the persistent recurrence launches leave `reserve_cus` CUs out of their residency arithmetic
    (yt8m_lstm_persist_reserve_cus): RCCL's kernels run beside the backward pass, and two half-chip recurrences admitted side by
    side would not both be resident next to them."""

    def __init__(self, group=None, bucket_bytes=32 << 20, overlap=True, comm=None, algo=None, reserve_cus=None):
        self.group = group
        self.comm = comm
        self.algo = algo or DP_ALGO
        if self.algo not in ("allreduce", "rs_ag"):
            raise ValueError("algo must be 'allreduce' or 'rs_ag'")
        self.reserve_cus = DP_RESERVED_CUS if reserve_cus is None else int(reserve_cus)
        self._prev_reserve = None
        self.world = comm.world if comm is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.overlap = overlap
        self.graph = None
        self._handles = []
        self._ready = None
        self._launched = None
        self.active = False

    def attach(self, graph, broadcast=True):
        """Hooks the graph and (broadcast=True) makes every rank start from rank 0's parameters.  The broadcast may write the arena
        through a raw pointer (CabiComm), which bumps no torch version counter: the resident weight images are rebuilt here, in
        stream order behind it.  A RE-attach after TrainGraph.close() passes broadcast=False: the ranks' parameters
        are already identical, and a broadcast between a step's forward and backward pass would change the weights under it."""
        self.graph = graph
        self.active = self.comm is not None or dist.is_initialized()
        if self.comm is not None:
            graph.rank = self.comm.rank
        else:
            graph.rank = dist.get_rank(self.group) if self.active else 0     # ranks draw different dropout / noise streams
        graph.grad_ready_hook = self._on_ready if (self.overlap and self.active) else None
        if broadcast and self.comm is not None:
            self.comm.broadcast(graph.params, 0)
        elif broadcast and self.active:
            dist.broadcast(graph.params, src=0, group=self.group)
        if broadcast and self.active and getattr(graph, "wimg", None) is not None:
            graph.wimg.refresh()
        nv = len(graph.trainable_variables())
        self._ready = [False] * nv
        self._launched = [False] * nv
        if self.active and self.reserve_cus > 0 and torch.cuda.is_available():
            import ctypes
            from . import _lib
            prev = ctypes.c_int(0)
            _lib.check(_lib.lib().yt8m_lstm_persist_reserve_cus(self.reserve_cus, ctypes.byref(prev)))
            self._prev_reserve = prev.value

    def detach(self):
        """Gives the CU headroom back (end of data-parallel training in this process)."""
        if self._prev_reserve is not None:
            from . import _lib
            _lib.check(_lib.lib().yt8m_lstm_persist_reserve_cus(self._prev_reserve, None))
            self._prev_reserve = None
        if self.graph is not None:
            self.graph.grad_ready_hook = None

    def begin_step(self):
        self._handles = []
        for i in range(len(self._ready)):
            self._ready[i] = False
            self._launched[i] = False
        if self.trace and torch.cuda.is_available():            # called right before final_loss.backward() (train.TrainGraph.step)
            self._t_start = torch.cuda.Event(enable_timing=True)
            self._t_start.record()
            self._t_buckets = []
            self._t_end = None

    # ---- tracing (bench.py --gpus N: "a SCALE line that comes back at 4x explains itself") ---------------------------------
    # With `trace` set, every bucket records when its collective was enqueued (stream order of the compute stream = the moment its
    # gradients were final) and when it landed (an event on a side stream that waited for the collective only), both relative to the
    # start of the backward pass; finished_buckets() marks the end of the backward pass.  Off in timed regions: the side stream is a
    # fifth active stream of the process and costs 0.1-1.7 ms of a headline step on its own.
    trace = False

    def _trace_launch(self, lo, hi, hs):
        if not (self.trace and torch.cuda.is_available()):
            return
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
        if getattr(self, "_trace_stream", None) is None:
            self._trace_stream = torch.cuda.Stream()
        ev1 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self._trace_stream):
            for h in hs:
                h.wait()
            ev1.record()
        self._t_buckets.append((lo, hi, ev0, ev1))

    def trace_report(self):
        """Timeline of the most recent traced step (synchronises): per bucket [element range, MB, enqueue / land in ms after the start
        of the backward pass], the length of the backward pass and the all-reduce time left exposed behind it."""
        if not getattr(self, "_t_buckets", None) or self._t_end is None:
            return None
        torch.cuda.synchronize()
        t0 = self._t_start
        bw = t0.elapsed_time(self._t_end)
        rows = [{"elements": [int(lo), int(hi)], "MB": (hi - lo) * 4 / 1e6, "enqueued_ms": t0.elapsed_time(e0), "landed_ms": t0.elapsed_time(e1)}
                for lo, hi, e0, e1 in self._t_buckets]
        last = max(r["landed_ms"] for r in rows)
        return {"backward_ms": bw, "buckets": rows, "exposed_allreduce_ms": max(0.0, last - bw),
                "note": "times relative to the start of the backward pass on this rank; exposed = last landing - end of backward"}

    def _span(self, i, j):
        tv = self.graph.trainable_variables()
        lo = tv[i].offset
        hi = tv[j].offset + tv[j].numel()
        return lo, hi

    def _on_ready(self, var):
        """Marks var ready; launches an async all-reduce for every maximal run of ready, unlaunched, adjacent
        variables whose size reaches the bucket threshold."""
        if self._launched[var.index]:
            # Variable.grad_done() fires this hook once, after the last of the variable's uses of this step (uses = get_variable
            # calls); a second report means an op contributed without fetching the variable through the graph: its gradient would
            # be added into a buffer whose all-reduce is already in flight: refuse loudly
            raise RuntimeError("gradient of %s was reported done twice in one step (variable shared between ops?): its "
                               "all-reduce is already in flight" % var.name)
        self._ready[var.index] = True
        self._flush(final=False)

    def _flush(self, final):
        n = len(self._ready)
        i = 0
        while i < n:
            if self._launched[i] or not (self._ready[i] or final):
                i += 1
                continue
            j = i
            while j + 1 < n and not self._launched[j + 1] and (self._ready[j + 1] or final):
                j += 1
            lo, hi = self._span(i, j)
            if final or hi - lo >= self.bucket_elems:
                # split very large runs so that several rings/links are in flight
                pos = lo
                hs = []
                while pos < hi:
                    end = min(hi, pos + 4 * self.bucket_elems)
                    if self.comm is not None:
                        hs.append(self.comm.all_reduce(self.graph.grads[pos:end], algo=self.algo))
                    elif self.algo == "rs_ag":
                        hs.append(_rsag_torch(self.graph.grads[pos:end], self.group))
                    else:
                        hs.append(dist.all_reduce(self.graph.grads[pos:end], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    pos = end
                self._trace_launch(lo, hi, hs)
                self._handles.append((i, j + 1, hs))      # trainable variables [i, j+1) ride on these handles
                for k in range(i, j + 1):
                    self._launched[k] = True
            i = j + 1

    def finished_buckets(self):
        """Yields (lo, hi) ranges of trainable-variable indices in the order their all-reduces were launched, each
        after waiting (stream-wise for RCCL) for that bucket only -- the caller runs clip+Adam on the bucket while later
        buckets are still on the wire.  Without an initialised process group: one bucket with everything."""
        n = len(self._ready) if self._ready is not None else len(self.graph.trainable_variables())
        if not self.active:
            yield (0, n)
            return
        if self.trace and torch.cuda.is_available() and getattr(self, "_t_end", 0) is None:
            self._t_end = torch.cuda.Event(enable_timing=True)  # the caller's backward() has returned: end of the pass in stream order
            self._t_end.record()
        self._flush(final=True)
        for lo, hi, hs in self._handles:
            for h in hs:
                h.wait()
            yield (lo, hi)
        self._handles = []

    @property
    def gscale(self):
        return 1.0 / self.world

    def finish(self):
        """Waits for every bucket; returns gscale = 1/world (the mean is folded into the optimiser pass)."""
        for _ in self.finished_buckets():
            pass
        return self.gscale
