"""World size 2 over gloo with the PRODUCT's node dispatcher doing the work (tests/test_sharding_gloo.py shards the
oracle, this shards through csrc/node.cpp).  Each rank builds nothing of its own: both load the library made of the real node.cpp
and the test-only single-GPU stand-ins (tests/node_mock/node_mock.cpp), create a node handle over three mock contexts, and take their
contiguous shard [n r / W, n (r + 1) / W) of ONE global batch (no data-path collective).

ACT_RNG_SEQUENTIAL must stay the byte stream of one sequential loop over one generator (src/lib.rs:638-643, 842-846: e and alpha are
drawn only for accepted lanes) ACROSS THE RANKS: every rank runs the check half on its shard (act_node_verify_spend_batch /
act_node_issue_check_batch: the dispatcher cuts the shard over its contexts), the accepted counts are gathered over gloo -- the only
communication, W integers -- every rank's offset into the stream is the number of accepted lanes in front of its shard, and the
sign half (act_node_refund_sign_batch / act_node_issue_sign_batch, inside which the dispatcher counts the accepted lanes in front of
each of ITS pieces again) starts there.  Rank 0 gathers the outputs and compares them with the sequential loop: lane i carries its
own record's tag and the rng slice number (accepted lanes before i)."""
import os
import random
import socket
import subprocess
import sys

import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PB = 64


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _records(n, rec, seed):
    r = random.Random(seed)
    out = bytearray()
    for i in range(n):
        out += bytes([r.randrange(256)]) + i.to_bytes(7, "little") + bytes(r.randrange(256) for _ in range(rec - 8))
    return bytes(out)


def _worker(rank, world, port, libpath, n, ret):
    import ctypes as C
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = C.CDLL(libpath)
    nd = C.c_void_p()
    devs = (C.c_int * 3)(0, 1, 2)
    assert lib.act_node_create(bytes(96), 128, devs, 3, C.c_size_t(0), C.byref(nd)) == 0
    proofs, reqs, rng = _records(n, PB, 11), _records(n, 128, 12), _records(n, 128, 13)      # the same global batch on every rank
    lo, hi = n * rank // world, n * (rank + 1) // world
    m = hi - lo
    sk = bytes(64)
    result = {}
    for what in ("refund", "issue"):
        st = C.create_string_buffer(max(1, m)); kp = C.create_string_buffer(max(1, 32 * m))
        if what == "refund":
            assert lib.act_node_verify_spend_batch(nd, C.c_size_t(m), sk, proofs[PB * lo:PB * hi] + b"\0", st, kp) == 0
        else:
            assert lib.act_node_issue_check_batch(nd, C.c_size_t(m), reqs[128 * lo:128 * hi] + b"\0", st) == 0
        accepted = sum(1 for i in range(m) if st.raw[i] == 0)
        counts = [None] * world
        dist.all_gather_object(counts, accepted)                  # the one exchange the path has: W integers
        before = sum(counts[:rank])
        total = sum(counts)
        rec = 128 if what == "refund" else 160
        out = C.create_string_buffer(max(1, rec * m)); st2 = C.create_string_buffer(max(1, m))
        stream = rng[128 * before:128 * (before + accepted)] + b"\0"                 # exactly this rank's part of the one stream
        if what == "refund":
            assert lib.act_node_refund_sign_batch(nd, C.c_size_t(m), sk, kp.raw[:32 * m] + b"\0", st.raw[:m] + b"\0", stream, 1, out, st2) == 0
        else:
            assert lib.act_node_issue_sign_batch(nd, C.c_size_t(m), sk, reqs[128 * lo:128 * hi] + b"\0", bytes(32 * m + 1), st.raw[:m] + b"\0", stream, 1, out, st2) == 0
        gathered = [None] * world
        dist.all_gather_object(gathered, (lo, st2.raw[:m], out.raw[:rec * m]))
        result[what] = (gathered, total)
    lib.act_node_ctx.restype = C.c_void_p; lib.act_node_ctx.argtypes = [C.c_void_p, C.c_int]
    lib.act_mock_lanes.restype = C.c_size_t; lib.act_mock_lanes.argtypes = [C.c_void_p]
    used = [lib.act_mock_lanes(lib.act_node_ctx(nd, k)) for k in range(3)]
    all_used = [None] * world
    dist.all_gather_object(all_used, used)
    if rank == 0:
        ret.put((result, all_used))
    dist.barrier()
    lib.act_node_destroy(nd)
    dist.destroy_process_group()


def test_two_ranks_shard_one_batch_through_the_product_dispatcher(tmp_path):
    libpath = str(tmp_path / "libnode_mock.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Werror", "-pthread", "-o", libpath,
                    os.path.join(ROOT, "anonymous-credit-tokens_amd", "csrc", "node.cpp"), os.path.join(ROOT, "tests", "node_mock", "node_mock.cpp")], check=True)
    n, world = 203, 2
    ctxmp = mp.get_context("spawn")
    q = ctxmp.Queue()
    port = _free_port()
    procs = [ctxmp.Process(target=_worker, args=(r, world, port, libpath, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    result, all_used = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    proofs, reqs, rng = _records(n, PB, 11), _records(n, 128, 12), _records(n, 128, 13)
    for what, inp, rec_in, rec_out, bad in (("refund", proofs, PB, 128, 7), ("issue", reqs, 128, 160, 1)):
        gathered, total = result[what]
        st = bytearray(n); out = bytearray(rec_out * n)
        covered = 0
        for lo, s, o in gathered:
            st[lo:lo + len(s)] = s; out[rec_out * lo:rec_out * lo + len(o)] = o; covered += len(s)
        assert covered == n                                         # every lane in exactly one rank's shard
        cur = 0                                                     # the sequential loop over ONE generator
        for i in range(n):
            acc = inp[rec_in * i] % 2 == 0
            assert st[i] == (0 if acc else bad), (what, i)
            got = bytes(out[rec_out * i:rec_out * (i + 1)])
            if not acc:
                assert got == bytes(rec_out), (what, i)
                continue
            assert got[:8] == inp[rec_in * i:rec_in * i + 8], "%s: lane %d carries another lane's record" % (what, i)
            assert got[8:16] == rng[128 * cur:128 * cur + 8], "%s: lane %d was handed slice %d, the loop hands it %d" % (what, i, int.from_bytes(got[9:16], "little"), cur)
            cur += 1
        assert cur == total
    # every mock context of both ranks did part of the work (the dispatcher really cut the shards)
    assert all(u > 0 for used in all_used for u in used), all_used
