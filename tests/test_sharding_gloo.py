"""N > 1 path of bench.py without GPUs: two gloo ranks shard independent batches (no data-path collective), time
them, and reduce the timing with MAX — the only communication the path has.  The 'work' here is the C oracle on a
handful of proofs so that the test exercises real disjoint shards."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from conftest import shake, scb
    from oracle_c import Oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle(); h = o.params_new("bench-org", "bench-service", "bench-env", "2024-01-01"); ctx = o.ctx(h, 8)
    sk = ctx.private_key_random(shake("sk", 64))
    # every rank derives the same global batch, then takes its contiguous shard
    lo, hi = n_total * rank // world, n_total * (rank + 1) // world
    proofs = b""
    for i in range(lo, hi):
        pre = ctx.pre_issuance_random(shake("pre%d" % i, 128)); req = ctx.request(pre, shake("rq%d" % i, 128))
        _, resp = ctx.issue(sk, req, scb(200), shake("ir%d" % i, 128)); _, tok = ctx.issuance_to_credit_token(pre, sk[32:], req, resp)
        _, p, _ = ctx.prove_spend(tok, scb(i if i != 3 else 201), shake("pr%d" % i, ctx.prove_rng_bytes))
        proofs += p
    dist.barrier()
    import time
    t0 = time.perf_counter(); st = ctx.verify_spend_batch(sk, proofs, 1); el = time.perf_counter() - t0
    t = torch.tensor([el + 0.01 * rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    counts = torch.tensor([len(st), sum(1 for s in st if s == 0)], dtype=torch.int64)
    dist.all_reduce(counts)          # host-side sum of accept counts: the only cross-shard value
    if rank == 0:
        ret.put((float(t.item()), el, counts.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_max_timing():
    ctxmp = mp.get_context("spawn")
    q = ctxmp.Queue()
    port = _free_port()
    procs = [ctxmp.Process(target=_worker, args=(r, 2, port, 6, q)) for r in range(2)]
    for p in procs:
        p.start()
    tmax, t0, counts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert counts == [6, 5]          # all lanes verified exactly once across shards; lane 3 overspends
    assert tmax >= t0
