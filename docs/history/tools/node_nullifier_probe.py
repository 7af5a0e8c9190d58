#!/usr/bin/env python3
"""Where a node-level nullifier call spends its time (act_node_nullifier_check_and_insert_batch from host memory): the host-side
routing by owner against the per-device look-ups, for 32-byte keys and for keys at the proof stride (what act_node_redeem_batch
passes), over 1 / 2 / 4 / 8 sets (all on device 0 on a one-GPU box: the host side is what is being measured)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from act_amd import capi

lib = capi.load()
n = 1 << int(os.environ.get("LOG2", "20"))
rng = np.random.default_rng(3)
for stride in (32, 16832):
    m = n if stride == 32 else n // 4
    buf = np.zeros(m * stride, np.uint8)
    buf.reshape(m, stride)[:, :32] = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    for parts in (1, 2, 4, 8):
        ns = capi.NodeNullifierSet(4 * m, devices=(0,) * parts)
        out = np.zeros(m, np.uint8)
        best = []
        for rep in range(3):
            # fresh keys every repetition: byte 31 of a key is cleared, byte 30 carries the repetition
            buf.reshape(m, stride)[:, 30] = rep; buf.reshape(m, stride)[:, 31] &= 0x0F
            t = time.perf_counter()
            rc = lib.act_node_nullifier_check_and_insert_batch(ns.h, m, buf.ctypes.data, stride, None, out.ctypes.data)
            best.append(time.perf_counter() - t)
            assert rc == 0 and not out.any()
        t = time.perf_counter()
        rc = lib.act_node_nullifier_check_and_insert_batch(ns.h, m, buf.ctypes.data, stride, None, out.ctypes.data)
        again = time.perf_counter() - t
        assert rc == 0 and out.all()
        print("stride %5d, %d set(s), %7d keys: insert %.1f ms (%.1f M keys/s), all-spent pass %.1f ms" % (stride, parts, m, 1e3 * min(best), m / min(best) / 1e6, 1e3 * again), flush=True)
        ns.close()
single = capi.NullifierSet(4 * n)
keys = rng.integers(0, 256, (n, 32), dtype=np.uint8); keys[:, 31] &= 0x0F
out = np.zeros(n, np.uint8)
t = time.perf_counter(); rc = lib.act_nullifier_check_and_insert_batch(single.h, n, capi.MEM_HOST, keys.ctypes.data, 32, None, out.ctypes.data); dt = time.perf_counter() - t
print("one per-device set, %d keys from host memory, no routing: %.1f ms" % (n, 1e3 * dt))
print("host pool:", capi.host_pool_stats())
