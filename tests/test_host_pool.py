"""The host side of the host-transcript mode (csrc/host_pool.cpp): ONE process-wide pool of BLAKE3 workers that every context of
the process takes a share of -- no thread creation per hash piece, fair shares between the contexts of a node handle.  Pure host
code: runs here without a GPU.  What is hashed is checked against the LLVM-BLAKE3 fixtures and against the one-thread path."""
import json
import os
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from act_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        capi.build()
    capi.load()
    return capi


def test_pool_hashes_upstream_vectors(capi):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "blake3_llvm.json")))
    for v in g["vectors"]:
        n = v["len"]
        if n > 20000:
            continue
        stride = (n + 64 + 15) & ~15
        count = 37                                  # two SIMD groups of sixteen + five through the scalar routine
        buf = np.zeros(stride * count, np.uint8)
        msg = np.frombuffer(bytes(j % 251 for j in range(n)), np.uint8)
        for i in range(count):
            buf[i * stride:i * stride + n] = msg
        out = capi.host_hash_many(buf, stride, n, count)
        for i in range(count):
            assert out[64 * i:64 * i + 64].hex() == v["xof"][:128], (n, i)


def test_eight_concurrent_callers_share_one_pool(capi):
    """Eight threads (the contexts of an 8-GPU node handle) hash spend-transcript-sized messages at the same time, four pieces
    each, as engine.hip hash_end does: same bytes as one thread alone, and the process creates its workers once."""
    length, stride, n = 15784, 15792, 700
    rng = np.random.default_rng(7)
    bufs = [rng.integers(0, 256, stride * n, dtype=np.uint8) for _ in range(8)]
    want = [capi.host_hash_many(b, stride, length, n, 1) for b in bufs]           # max_threads = 1: the caller alone
    before = capi.host_pool_stats()
    got = [None] * 8

    def ctx(k):
        pieces = []
        for p in range(4):
            i0, i1 = n * p // 4, n * (p + 1) // 4
            pieces.append(capi.host_hash_many(bufs[k][i0 * stride:], stride, length, i1 - i0))
        got[k] = b"".join(pieces)
    for _ in range(3):
        th = [threading.Thread(target=ctx, args=(k,)) for k in range(8)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert got == want
    after = capi.host_pool_stats()
    assert after["jobs"] - before["jobs"] == 3 * 8 * 4
    assert after["pool_size"] == capi.host_usable_cpus() >= 1
    assert after["threads_created"] <= after["pool_size"] - 1 + 0        # one set of workers for the whole process ...
    # ... and hashing more creates none
    capi.host_hash_many(bufs[0], stride, length, n)
    assert capi.host_pool_stats()["threads_created"] == after["threads_created"]


def test_edge_sizes(capi):
    assert capi.host_hash_many(np.zeros(16, np.uint8), 16, 3, 0) == b""
    one = capi.host_hash_many(np.zeros(64, np.uint8), 64, 0, 1)
    assert one[:32].hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"      # BLAKE3("")


def test_parallel_for_covers_every_index_once(capi):
    """act_host_parallel_for (what the node-level nullifier set routes keys with): every index in exactly one item, items no
    longer than the grain, callers on several threads at the same time, nothing lost."""
    import ctypes as C
    lib = capi.load()
    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t)

    def run(n, grain, par):
        hits = np.zeros(n, np.int32); sizes = []; lock = threading.Lock()

        def item(_, i0, i1):
            hits[i0:i1] += 1                       # items never overlap, so no two threads write one element
            with lock:
                sizes.append(i1 - i0)
        lib.act_host_parallel_for(n, grain, par, CB(item), None)
        assert (hits == 1).all() and sum(sizes) == n and max(sizes) <= max(1, grain)

    for n, grain, par in ((1, 1, 0), (1000, 7, 0), (4096, 64, 3), (5, 100, 0), (257, 1, 1)):
        run(n, grain, par)
    lib.act_host_parallel_for(0, 1, 0, CB(lambda *_: None), None)
    th = [threading.Thread(target=run, args=(3000 + k, 11, 0)) for k in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
