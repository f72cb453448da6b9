"""N > 1 path on CPU: world-size-2 gloo processes exercise the pair partition, the throughput reduction, the
result gather, the gradient all-reduce and the hypothesis-split merge helpers (no GPU compute: the hot path itself has no CPU fallback)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from differentiable_ransac_amd import sharding, synth


def test_pair_range_partitions_everything():
    for P in (1, 2, 7, 32, 255, 256):
        for G in (1, 2, 3, 8):
            spans = [sharding.pair_range(P, r, G) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == P
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = 7
        batch = synth.batch_two_view(P, 16, seed0=3)
        mine = sharding.shard_pairs(batch, rank, world)
        lo, hi = sharding.pair_range(P, rank, world)
        assert mine["matches"].shape[0] == hi - lo
        assert torch.equal(mine["matches"], batch["matches"][lo:hi])
        # throughput: rank 0 "runs" 1 s, rank 1 2 s
        hyps, secs = sharding.job_throughput((hi - lo) * 1024, 1.0 + rank, dist)
        assert secs == 2.0 and abs(hyps - P * 1024 / 2.0) < 1e-9
        # a per-pair result (here: the pair's first match) gathered back in order
        got = sharding.gather_results(mine["matches"][:, 0], P, dist)
        assert torch.equal(got, batch["matches"][:, 0])
        # gradient all-reduce (mean)
        g = [torch.full((3, 2), float(rank + 1)), torch.full((5,), 10.0 * (rank + 1))]
        sharding.allreduce_mean_(g, dist)
        assert torch.allclose(g[0], torch.full((3, 2), 1.5)) and torch.allclose(g[1], torch.full((5,), 15.0))
        # hypothesis split (P < G): both ranks hold a local best for the SAME pairs; the merge keeps the better one,
        # ties go to the lowest rank, NaN scores never win
        sc = torch.tensor([[3.0, 1.0, 5.0, float("nan")], [2.0, 4.0, 5.0, 0.5]])[rank]
        md = torch.full((4, 3, 3), float(rank))
        inl = torch.tensor([[30, 10, 50, 0], [20, 40, 51, 5]], dtype=torch.int32)[rank]
        s, m, w, i = sharding.merge_best(sc, md, (inl,), dist)
        assert w.tolist() == [0, 1, 0, 1]
        assert s.tolist() == [3.0, 4.0, 5.0, 0.5] and i.tolist() == [30, 40, 50, 5]
        assert torch.equal(m[:, 0, 0], w.float())
        assert sharding.hypothesis_seed(7, 0) == 7 and sharding.hypothesis_seed(7, 1) != sharding.hypothesis_seed(7, 2)
        # the training step's collective as bench.py --mode train issues it (N > 1): ONE flat bucket = the scores network's
        # gradient (622 616 f32, the reference CLNet).  The per-pair logits gradient is NOT in it: ranks own different pairs, it
        # continues into each rank's own backward through the network
        net = torch.full((622616,), float(rank + 1))
        sharding.allreduce_mean_([net], dist)
        assert float(net[0]) == 1.5 and float(net[-1]) == 1.5
        # ... and asynchronously, overlapped with the next step: step i + 1's launches are enqueued BEFORE the wait on bucket i
        bucket = sharding.AsyncGradientBucket(622616, "cpu", dist)
        seen = []

        def fake_step():                       # stands for the graph replay of forward + backward: fills the bucket to be reduced
            bucket.bucket().fill_(float((rank + 1) * (len(seen) + 1)))
            seen.append(len(seen))
        step = sharding.OverlappedStep(fake_step, bucket)
        for _ in range(4):
            step()
        bucket.drain()
        tr = bucket.trace
        for i in range(3):
            assert tr.index(("enqueued", i + 1)) < tr.index(("wait", i)) < tr.index(("launch", i + 1)), tr
        assert tr.index(("launch", 0)) < tr.index(("enqueued", 1)) and ("wait", 3) in tr
        # the averaged values: bucket i held (rank + 1) * (i + 1) -> mean over two ranks = 1.5 (i + 1); buffers alternate
        assert float(bucket.buf[1][0]) == 1.5 * 4 and float(bucket.buf[0][-1]) == 1.5 * 3
        one = torch.ones(1)
        dist.all_reduce(one)                       # bench.py's n_ranks_seen
        assert int(one.item()) == world
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_world_size_two_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _worker8(rank, world, port, q):
    """BASELINE configs[4] at its own world size: 256 pairs -> 32 per rank, one flat bucket of 622 616 f32, overlapped"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = 256
        lo, hi = sharding.pair_range(P, rank, world)
        assert hi - lo == 32 and lo == 32 * rank
        # every pair is owned exactly once: gather the owners of the job's pairs
        own = torch.zeros(P, dtype=torch.int64)
        own[lo:hi] = 1
        dist.all_reduce(own)
        assert own.tolist() == [1] * P
        # a per-pair result (the pair's index, computed on the owning rank) gathered back in pair order
        got = sharding.gather_results(torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 9), P, dist)
        assert got.shape == (P, 9) and torch.equal(got[:, 0], torch.arange(P, dtype=torch.float32))
        # whole-job throughput: sum of the ranks' hypotheses over the slowest rank's time
        hyps, secs = sharding.job_throughput(32 * 1024, 1.0 + 0.1 * rank, dist)
        assert abs(secs - 1.7) < 1e-12 and abs(hyps - 256 * 1024 / 1.7) < 1e-6
        # hypothesis split over 8 ranks (P < G): best score per pair over the ranks, ties to the lowest rank, NaN never wins
        sc = torch.tensor([float(rank), float(7 - rank), 3.0, float("nan") if rank != 5 else 0.25])
        md = torch.full((4, 3, 3), float(rank))
        s, m, w = sharding.merge_best(sc, md, (), dist)[:3]
        assert w.tolist() == [7, 0, 0, 5] and s.tolist() == [7.0, 7.0, 3.0, 0.25]
        assert torch.equal(m[:, 0, 0], w.float())
        assert len({sharding.hypothesis_seed(11, r) for r in range(world)}) == world
        # the train step's bucket, asynchronously, over 8 ranks: mean of (rank + 1) * (i + 1) over the ranks = 4.5 (i + 1)
        bucket = sharding.AsyncGradientBucket(622616, "cpu", dist)
        n = [0]

        def fake_step():
            bucket.bucket().fill_(float((rank + 1) * (n[0] + 1)))
            n[0] += 1
        reduced = []
        step = sharding.OverlappedStep(fake_step, bucket, on_reduced=lambda t, i: reduced.append((i, float(t[0]), float(t[-1]))))
        for _ in range(3):
            step()
        bucket.drain()
        tr = bucket.trace
        for i in range(2):
            assert tr.index(("enqueued", i + 1)) < tr.index(("wait", i)) < tr.index(("launch", i + 1)), tr
        assert reduced == [(0, 4.5, 4.5), (1, 9.0, 9.0)], reduced
        assert float(bucket.buf[0][0]) == 13.5
        one = torch.ones(1)
        dist.all_reduce(one)
        assert int(one.item()) == world
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_world_size_eight_gloo():
    """the target world size of BASELINE configs[4] (8 ranks x 32 pairs) on CPU processes"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(8)], res
