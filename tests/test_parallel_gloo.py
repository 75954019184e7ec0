"""world_size-2 `gloo` tests (CPU) of the data-parallel path: parameter broadcast, bucketed / overlapped gradient
all-reduce over the arena, equal sharding, and the SURVEY.md 8e identity "N-rank step on shards == 1-rank step on
the global batch" (checked with the CPU restatement: mean of per-shard gradients == global-batch gradient)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, overlap, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        __graft_entry__.load_package()
        from yt8m_amd import parallel
        from yt8m_amd.variables import Graph, random_normal, zeros
        from oracle import torch_ref
        r, w, _ = parallel.init_from_env(backend="gloo")
        assert (r, w) == (rank, world)
        g = Graph(device="cpu", seed=100 + rank)               # different init per rank on purpose
        g.begin_step()
        shapes = {"gates/weights": (40, 90), "experts/weights": (40, 60), "experts/biases": (60,), "tiny": (3,)}
        vs = {k: g.get_variable(k, s, random_normal(0.1)) for k, s in shapes.items()}
        g.finalize()
        # (the non-overlapped run also takes the reduce-scatter + all-gather form: on gloo, which has no reduce-scatter, it must
        # fall back to one all-reduce per bucket and give the same sums)
        red = parallel.GradReducer(bucket_bytes=4 * 2000, overlap=overlap, algo="allreduce" if overlap else "rs_ag")
        assert red.algo == ("allreduce" if overlap else "rs_ag")
        red.attach(g)
        # 0. the random-op stream (dropout / noise keys) is rank-dependent and reproducible from (seed, rank, pass, call)
        from yt8m_amd.variables import random_seed
        assert g.rank == rank
        g.begin_step()
        k0, k1 = g.next_random_seed(), g.next_random_seed()
        assert k0 == random_seed(100 + rank, rank, g._rng_step, 0) and k1 == random_seed(100 + rank, rank, g._rng_step, 1) and k0 != k1
        keys = [None] * world
        import torch.distributed as dist
        dist.all_gather_object(keys, random_seed(7, rank, 3, 0))
        assert len(set(keys)) == world                      # same graph seed, same pass, same call: ranks draw different masks
        # 1. broadcast: every rank now holds rank 0's parameters
        ref = Graph(device="cpu", seed=100)
        ref.begin_step()
        for k, s in shapes.items():
            ref.get_variable(k, s, random_normal(0.1))
        for k in shapes:
            assert torch.equal(vs[k].data, ref.vars[k].data), k
        # 2. all-reduce with grad-ready hooks fired in reverse creation order (as backward does)
        for step in range(2):
            red.begin_step()
            for k in reversed(list(shapes)):
                vs[k].grad.fill_(float(rank + 1) * (step + 1))
                vs[k].grad_done()
            order = list(red.finished_buckets())            # buckets complete in launch order, covering every variable once
            assert sorted(i for lo, hi in order for i in range(lo, hi)) == list(range(len(shapes)))
            assert len(order) >= 2 if overlap else len(order) == 1
            gscale = red.gscale
            assert gscale == 0.5
            for k in shapes:
                assert torch.all(vs[k].grad == 3.0 * (step + 1)), (k, vs[k].grad.flatten()[:3])
        # 3. DP identity on the real math (CPU restatement): mean over ranks of shard gradients == global gradient
        torch.manual_seed(0)
        B, Dm, V, M = 8, 12, 10, 2
        x = torch.randn(B, Dm, dtype=torch.float64)
        y = torch.rand(B, V) < 0.3
        P = torch_ref.make_moe_params(Dm, V, M, torch.float64, seed=1)
        lo, hi = parallel.shard_batch(B, rank, world)

        def grads(xs, ys):
            Pl = {k: v.clone().requires_grad_(True) for k, v in P.items()}
            torch_ref.cross_entropy(torch_ref.moe(torch_ref.l2_normalize(xs, 1), Pl["gates/weights"], Pl["experts/weights"],
                                                  Pl["experts/biases"], M), ys).backward()
            return torch.cat([Pl[k].grad.reshape(-1) for k in sorted(Pl)])
        gl = grads(x[lo:hi], y[lo:hi])
        dist.all_reduce(gl)
        gl *= gscale
        assert torch.allclose(gl, grads(x, y), atol=1e-12)
        with pytest.raises(ValueError):
            parallel.shard_batch(7, rank, world)
        # 4. distributed evaluation: each rank contributes the top-k pairs of ITS shard (uneven shards on
        #    purpose); every rank ends up with the GAP / Hit@1 / loss of the global batch
        import numpy as np
        import yt8m_amd.eval_util as eval_util
        rs = np.random.RandomState(5)
        Bg, Vg, k = 11, 30, 5
        pg = rs.rand(Bg, Vg).astype(np.float32)
        yg = rs.rand(Bg, Vg) < 0.2
        cut = 4
        sl = slice(0, cut) if rank == 0 else slice(cut, Bg)
        pt, yt = torch.from_numpy(pg[sl]), torch.from_numpy(yg[sl]).float()
        vals, idx = torch.topk(pt, k, dim=1)                      # (test-side top-k: the product's runs on the device)
        em = eval_util.EvaluationMetrics(Vg, k)
        perr = torch.tensor(eval_util.calculate_precision_at_equal_recall_rate(pg[sl], yg[sl]) * pt.shape[0], dtype=torch.float64)
        out = em.accumulate_topk(vals, torch.gather(yt, 1, idx), idx, yt.sum(0), perr, loss=float(rank + 1))
        ref = eval_util.EvaluationMetrics(Vg, k)
        ref.accumulate(pg, yg, np.array([(1.0 * cut + 2.0 * (Bg - cut)) / Bg]))
        got, exp = em.get(), ref.get()
        assert abs(got["gap"] - exp["gap"]) < 1e-12 and abs(got["avg_hit_at_one"] - exp["avg_hit_at_one"]) < 1e-12
        assert abs(got["avg_loss"] - exp["avg_loss"]) < 1e-12 and em.num_examples == Bg
        assert abs(out["hit_at_one"] - exp["avg_hit_at_one"]) < 1e-12
        assert abs(got["avg_perr"] - exp["avg_perr"]) < 1e-12            # all five keys agree with the host path
        assert np.allclose(got["aps"], exp["aps"], atol=1e-12)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_gloo(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_reducer_is_identity():
    __graft_entry__.load_package()
    from yt8m_amd import parallel
    from yt8m_amd.variables import Graph, zeros
    g = Graph(device="cpu")
    g.begin_step()
    v = g.get_variable("w", (4,), zeros)
    g.finalize()
    red = parallel.GradReducer()
    red.attach(g)
    red.begin_step()
    v.grad.fill_(2.0)
    v.grad_done()
    assert red.finish() == 1.0 and torch.all(v.grad == 2.0)
    assert parallel.shard_batch(8, 1, 4) == (2, 4)


def test_grad_done_fires_after_the_last_use_of_a_shared_variable():
    """A variable fetched twice in one forward pass (get_variable is get-or-create) has two gradient contributions: the
    graph's gradient-complete hook -- the reducer's all-reduce trigger -- fires on the second grad_done(), not the first, and
    the second contribution accumulates (beta = 1) instead of raising (shared variables under data parallelism)."""
    __graft_entry__.load_package()
    from yt8m_amd.variables import Graph, zeros
    g = Graph(device="cpu")
    g.begin_step()
    a = g.get_variable("shared", (3,), zeros)
    b = g.get_variable("single", (2,), zeros)
    g.finalize()
    fired = []
    g.grad_ready_hook = lambda v: fired.append(v.name)
    for step in range(2):                                   # the per-step counters reset in begin_step
        g.begin_step()
        assert g.get_variable("shared", (3,), zeros) is a and g.get_variable("shared", (3,), zeros) is a
        assert g.get_variable("single", (2,), zeros) is b
        fired.clear()
        assert a.grad_beta() == 0.0
        a.grad_done()
        assert fired == []                                  # one of two uses reported
        assert a.grad_beta() == 1.0                         # the second contribution accumulates, no error
        a.grad_done()
        assert fired == ["shared"]
        assert b.grad_beta() == 0.0
        b.grad_done()
        assert fired == ["shared", "single"]
        with pytest.raises(RuntimeError):                   # a third contribution after the hook fired is an error
            a.grad_beta()
