"""Host BLAKE3 (csrc/host_hash.cpp, 16 messages per call) throughput on this box's cores, at the spend transcript size."""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from act_amd import capi
lib = C.CDLL(capi.LIB_PATH)
lib.act_host_b3_xof64_x16.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
LEN, STRIDE, N = 15784, 15792, 4096
buf = np.random.randint(0, 256, N * STRIDE, dtype=np.uint8)
out = np.zeros(N * 16, np.uint32)
def work(lo, hi):
    for i in range(lo, hi, 16):
        lib.act_host_b3_xof64_x16(buf.ctypes.data + i * STRIDE, STRIDE, LEN, out.ctypes.data + i * 64)
for nt in (1, 8, 16, 32):
    th = [threading.Thread(target=work, args=(N * t // nt // 16 * 16, N * (t + 1) // nt // 16 * 16)) for t in range(nt)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    print("%2d threads: %.2f GB/s (%.0f transcripts/s)" % (nt, N * LEN / dt / 1e9, N / dt))
