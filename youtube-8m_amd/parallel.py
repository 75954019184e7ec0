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
# CU headroom of the persistent recurrences for RCCL's kernels.  A reserve of R CUs makes the half-chip backward recurrences of the two
# layers run one at a time instead of side by side (they no longer both fit): +2.0 ms on the 24 ms step of round 3, +2.7 ms on round 6's
# 16 ms step (profiles/r6_force_reducer.md: 20.8 against 18.0 ms with an 8-GPU all-reduce's footprint emulated).  Without a reserve a recurrence launched while a collective's kernels hold CUs spins on the
# part of its grid that is resident until they leave -- bounded, never a deadlock (the collective never waits for a recurrence;
# tests/test_gpu_round3.py::test_persistent_recurrences_next_to_a_cu_hogging_kernel) -- so the price of NO reserve is the time a
# collective is still on the wire when the side-by-side phase of the backward pass begins.
#   YT8M_DP_RESERVED_CUS = <int>  : that many CUs, always
#   YT8M_DP_RESERVED_CUS = auto   : (default) chosen by auto_reserve_cus() from the size of the first large bucket, the world size and an
#                                   assumed bus bandwidth: the reserve is taken only when the collision it avoids costs more than it does
_RES = os.environ.get("YT8M_DP_RESERVED_CUS", "auto")
DP_RESERVED_CUS = None if _RES == "auto" else int(_RES)
DP_BUSBW_GBPS = float(os.environ.get("YT8M_DP_BUSBW_GBPS", "300"))       # all-reduce bus bandwidth assumed by the auto rule (8 x MI355X over
                                                                          # xGMI; no N > 1 measurement of this build exists: an assumption)
# measured on one MI355X (profiles/r6_force_reducer.md): the head's bucket is enqueued ~0.3 ms into the backward pass and the first
# side-by-side pair of recurrences starts ~1.9 ms later; chaining the recurrences (any reserve > 0) costs ~2.7 ms per step
DP_LONE_WINDOW_MS = float(os.environ.get("YT8M_DP_LONE_WINDOW_MS", "1.9"))
DP_RESERVE_COST_MS = float(os.environ.get("YT8M_DP_RESERVE_COST_MS", "2.7"))
DP_RCCL_CUS = int(os.environ.get("YT8M_DP_RCCL_CUS", "32"))              # CUs an RCCL all-reduce of a large message holds (channels)


def allreduce_ms(nbytes, world, busbw_gbps=None):
    """Ring all-reduce time of nbytes on `world` ranks at a bus bandwidth: 2 (N - 1) / N x bytes / busbw."""
    if world <= 1:
        return 0.0
    return 2.0 * (world - 1) / world * nbytes / ((busbw_gbps or DP_BUSBW_GBPS) * 1e9) * 1e3


def auto_reserve_cus(world, early_bucket_bytes, persistent_recurrences=True):
    """CUs the persistent recurrences should leave to RCCL.  0 at world 1 and for graphs without persistent recurrences.  Otherwise
    the collective of the gradients that are final when the backward pass of the recurrent stack begins (the head: `early_bucket_bytes`)
    runs beside the first, lone, half-chip recurrence; what is still on the wire when the two layers' recurrences start running side
    by side delays one of them by that remainder.  Reserve only if that remainder exceeds what chaining the recurrences costs."""
    if world <= 1 or not persistent_recurrences:
        return 0
    collision_ms = max(0.0, allreduce_ms(early_bucket_bytes, world) - DP_LONE_WINDOW_MS)
    return DP_RCCL_CUS if collision_ms > DP_RESERVE_COST_MS else 0


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
        self.reserve_cus = DP_RESERVED_CUS if reserve_cus is None else int(reserve_cus)     # None: chosen in attach() (auto rule)
        self.reserve_rule = None
        self._prev_reserve = None
        # YT8M_DP_EMULATE="cus:world[:busbw]" (bench.py --force-reducer on ONE GPU): every bucket's collective is followed on the
        # collective's timeline by a kernel that holds `cus` CUs for the time a `world`-rank ring all-reduce of the bucket would take
        # at the bus bandwidth -- the footprint a 1-rank RCCL group does not have.  Measurement aid only (profiles/r6_force_reducer.md).
        self.emulate = None
        emu = os.environ.get("YT8M_DP_EMULATE")
        if emu:
            f = emu.split(":")
            self.emulate = (int(f[0]), int(f[1]), float(f[2]) if len(f) > 2 else DP_BUSBW_GBPS)
        self._emu_stream = self._emu_out = None
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
        if self.reserve_cus is None:
            # auto: from the gradients that are final when a recurrent stack's backward pass starts (everything behind the last
            # recurrent variable in creation order = the head); graphs without persistent recurrences get 0
            tv = graph.trainable_variables()
            rec = [i for i, v in enumerate(tv) if "lstm_cell" in v.name or "/RNN/" in v.name or v.name.startswith("RNN/")]
            early = sum(v.numel() * 4 for v in tv[max(rec) + 1:]) if rec else 0
            w_eff = self.emulate[1] if (self.emulate and self.world == 1) else self.world
            self.reserve_cus = auto_reserve_cus(w_eff if self.active else 1, early, bool(rec))
            self.reserve_rule = {"rule": "auto", "early_bucket_MB": early / 1e6, "world": w_eff, "busbw_GBps": DP_BUSBW_GBPS,
                                 "allreduce_ms": allreduce_ms(early, w_eff), "lone_window_ms": DP_LONE_WINDOW_MS,
                                 "reserve_cost_ms": DP_RESERVE_COST_MS, "chosen": self.reserve_cus}
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
                if self.emulate is not None and torch.cuda.is_available():
                    hs.append(self._emulate_footprint((hi - lo) * 4, hs))
                self._trace_launch(lo, hi, hs)
                self._handles.append((i, j + 1, hs))      # trainable variables [i, j+1) ride on these handles
                for k in range(i, j + 1):
                    self._launched[k] = True
            i = j + 1

    def _emulate_footprint(self, nbytes, hs):
        """A kernel holding emulate[0] CUs for the ring all-reduce time of nbytes on emulate[1] ranks, behind the bucket's real
        (1-rank) collective on a side stream; returns a handle the optimiser pass of the bucket waits for."""
        import ctypes
        from . import _lib
        cus, world, busbw = self.emulate
        if self._emu_stream is None:
            self._emu_stream = torch.cuda.Stream()
            self._emu_out = torch.zeros(2 * 1024, dtype=torch.int32, device="cuda")
        ms = allreduce_ms(nbytes, world, busbw)
        with torch.cuda.stream(self._emu_stream):
            for h in hs:
                h.wait()                                         # behind the real collective (stream order of the side stream)
            _lib.check(_lib.lib().yt8m_probe_placement(ctypes.c_void_p(self._emu_out.data_ptr()), int(cus), int(ms * 1e5),
                                                       ctypes.c_void_p(self._emu_stream.cuda_stream)))     # wall clock: 100 MHz
            ev = torch.cuda.Event()
            ev.record(self._emu_stream)
        return _CabiHandle(ev)

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
