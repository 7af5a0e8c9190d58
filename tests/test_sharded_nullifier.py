"""ShardedNullifierSet (anonymous-credit-tokens_amd/sharded_nullifier.py): the key-space partition + all-to-all around
the per-GPU nullifier set, against the sequential meaning of the reference tests' NullifierDb
(/root/reference/src/tests.rs:29-50) applied to the concatenation of all ranks' batches in rank order.

CPU part: two gloo ranks, CPU tensors, and a dict-backed stand-in for the local shard (test-only: the product's local
shard is the HIP set and nothing else).  GPU part: the real HIP set behind a one-rank RCCL group, so the exchange code
runs on device tensors through the C ABI."""
import hashlib
import os
import random
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _key(i: int) -> bytes:
    return hashlib.shake_256(b"nul%d" % i).digest(32)


def _batches(world: int, rounds: int):
    """Deterministic per-(round, rank) key lists with repeats inside a batch, across ranks and across rounds; ragged sizes
    including an empty batch; a skip mask on odd rounds."""
    r = random.Random(11)
    out = []
    for rnd in range(rounds):
        per_rank = []
        for rank in range(world):
            n = [0, 1, 37, 200, 513][(rnd + 2 * rank) % 5]
            ids = [r.randrange(300) for _ in range(n)]
            mask = [1 if (rnd % 2 and r.random() < 0.15) else 0 for _ in range(n)]
            per_rank.append((ids, mask))
        out.append(per_rank)
    return out


def _sequential(world: int, rounds):
    """The loop of src/tests.rs:29-50 over all ranks' batches in (round, rank, lane) order."""
    db, exp = set(), []
    for per_rank in rounds:
        row = []
        for ids, mask in per_rank:
            flags = []
            for i, m in zip(ids, mask):
                if m:
                    flags.append(0)
                elif i in db:
                    flags.append(1)
                else:
                    db.add(i); flags.append(0)
            row.append(flags)
        exp.append(row)
    return exp, len(db)


class _DictShard:
    """Test-only local shard with the lane-order semantics of act_nullifier_check_and_insert_batch."""

    def __init__(self):
        self.db = set()

    def check_and_insert_tensor(self, keys):
        out = torch.zeros(keys.shape[0], dtype=torch.uint8)
        for i in range(keys.shape[0]):
            k = bytes(keys[i].tolist())
            if k in self.db:
                out[i] = 1
            else:
                self.db.add(k)
        return out

    def __len__(self):
        return len(self.db)


def _run(s, rounds, rank, device):
    got = []
    for per_rank in rounds:
        ids, mask = per_rank[rank]
        keys = torch.tensor([list(_key(i)) for i in ids], dtype=torch.uint8).reshape(len(ids), 32).to(device)
        m = torch.tensor(mask, dtype=torch.uint8).to(device) if any(mask) else None
        got.append(s.check_and_insert(keys, m).cpu().tolist())
    return got


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from act_amd.sharded_nullifier import ShardedNullifierSet
    shard = _DictShard()
    s = ShardedNullifierSet(0, local_set=shard)
    rounds = _batches(world, 7)
    got = _run(s, rounds, rank, "cpu")
    # every key this shard holds is owned by this rank
    own_ok = all(int.from_bytes(k[:8], "little") % (1 << 63) % world == rank for k in shard.db)
    q.put((rank, got, len(shard), own_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_matches_the_sequential_set():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(world):
        rank, got, held, own_ok = q.get(timeout=180)
        res[rank] = (got, held, own_ok)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rounds = _batches(world, 7)
    exp, total = _sequential(world, rounds)
    for rank in range(world):
        got, held, own_ok = res[rank]
        assert own_ok
        for rnd in range(len(rounds)):
            assert got[rnd] == exp[rnd][rank], (rank, rnd)
    assert sum(res[r][1] for r in range(world)) == total
    assert all(res[r][1] > 0 for r in range(world))          # the key space really is split


@pytest.mark.gpu
def test_hip_set_behind_a_one_rank_rccl_group():
    sys.path.insert(0, ROOT)
    from act_amd import capi            # noqa: F401  (loads the HIP library after torch)
    from act_amd.sharded_nullifier import ShardedNullifierSet
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        s = ShardedNullifierSet(100_000, device=0)
        rounds = _batches(1, 7)
        exp, total = _sequential(1, rounds)
        got = _run(s, rounds, 0, "cuda")
        for rnd in range(len(rounds)):
            assert got[rnd] == exp[rnd][0], rnd
        assert len(s) == total
        s.close()
    finally:
        dist.destroy_process_group()


def test_keys_are_reduced_before_routing():
    """k and k + l (the verifier accepts both spellings) must go to one owner as one key: reduce_mod_l against Python integers,
    incl. the values whose (x mod 2^252) - q c goes negative."""
    import act_amd  # noqa: F401
    from act_amd.sharded_nullifier import reduce_mod_l, _ELL
    r = random.Random(3)
    vals = [0, 1, _ELL - 1, _ELL, _ELL + 1, 2 * _ELL - 1, 15 * _ELL, 15 * _ELL + 5, 2**256 - 1, 2**252, 2**252 + 5, 2**253, 3 * 2**252 + 7,
            15 * 2**252, 15 * 2**252 + (_ELL - 2**252) * 15 - 1] + [r.randrange(2**256) for _ in range(500)]
    t = torch.tensor([list(v.to_bytes(32, "little")) for v in vals], dtype=torch.uint8)
    out = reduce_mod_l(t)
    for v, row in zip(vals, out.tolist()):
        assert int.from_bytes(bytes(row), "little") == v % _ELL, hex(v)
    assert reduce_mod_l(t[:0]).shape == (0, 32)
