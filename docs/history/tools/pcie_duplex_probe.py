#!/usr/bin/env python3
"""Does this box move host -> device and device -> host copies at the same time?  Two streams, pinned buffers: each direction alone,
then both at once.  (The codec's host-memory path alternates its chunks between two streams to use both directions.)"""
import time
import torch

n = 1 << 30
h_in = torch.empty(n, dtype=torch.uint8, pin_memory=True); h_out = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d_a = torch.empty(n, dtype=torch.uint8, device="cuda"); d_b = torch.zeros(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(up, down, reps=4):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        if up:
            with torch.cuda.stream(s1):
                d_a.copy_(h_in, non_blocking=True)
        if down:
            with torch.cuda.stream(s2):
                h_out.copy_(d_b, non_blocking=True)
    torch.cuda.synchronize()
    return reps * n / (time.perf_counter() - t) / 1e9


run(True, True, 1)
print("host -> device alone: %.1f GB/s" % run(True, False))
print("device -> host alone: %.1f GB/s" % run(False, True))
both = run(True, True)
print("both at once: %.1f GB/s each way (%.1f GB/s in total)" % (both, 2 * both))
