#!/usr/bin/env python3
"""bench.py — spend-proof verifies/sec (whole node), batch 2^20 per GPU, L = 128 (BASELINE.json metric).

A step = one pass of the hot path (PrivateKey::refund up to the challenge check, src/lib.rs:787-844) over one
batch of 2^20 synthetic spend proofs that are already resident in HBM when the timed region starts; transcripts
are hashed by the device BLAKE3 kernel (byte-identical to the host path; the host-transcript rate is printed as
an extra key).  One process per GPU; ranks shard independent batches (weak scaling, no data-path collective —
torch.distributed/RCCL is used only for the barrier and the max-over-ranks reduction of the timing).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ELL = 2**252 + 27742317777372353535851937790883648493
L = 128
PROOF_BYTES = 32 * (14 + 4 * L)          # 16 832 (algorithmic bytes in per verify)
ALGO_BYTES_PER_VERIFY = PROOF_BYTES + 1  # + 1 status byte out


def shake(label, n):
    return hashlib.shake_256(label.encode()).digest(n)


def scb(v):
    return (v % ELL).to_bytes(32, "little")


def make_inputs(eng, sk, distinct):
    """`distinct` valid spend proofs made by the engine itself (request -> issue -> token -> prove_spend):
    bench Params, c uniform in [20,1000), s uniform in [1,c-1] (benches/benchmark.rs:131,147-154)."""
    import random
    r = random.Random(20240101)
    pre = eng.pre_issuance_random(shake("bench-pre", 128 * distinct))
    req = eng.request(pre, shake("bench-rq", 128 * distinct))
    cs = [r.randrange(20, 1000) for _ in range(distinct)]
    st, resp = eng.issue(sk, req, b"".join(scb(c) for c in cs), shake("bench-ir", 128 * distinct))
    assert st == bytes(distinct)
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    assert st == bytes(distinct)
    ss = [r.randrange(1, c) for c in cs]
    st, proofs, _ = eng.prove_spend(tok, b"".join(scb(s) for s in ss), shake("bench-pr", eng.prove_rng_bytes * distinct))
    assert st == bytes(distinct)
    return proofs


def usable_cores():
    """Host threads this process may actually use: CPU affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(proofs_host, h, sk, seconds_target=12.0):
    """The C oracle (a restatement of the reference algorithm with the reference's operation structure — NOT the
    Rust crate, which cannot be built here) timed on this box's host cores on a bounded sample of the same proofs."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    native = os.path.join("/tmp", "libact_oracle_native_%d.so" % os.getpid())
    try:
        oracle_c.build(native_out=native)
        o = oracle_c.Oracle(native)
    except Exception:
        o = oracle_c.Oracle()
    ctx = o.ctx(h, L)
    cores = usable_cores()
    t = time.perf_counter(); st = ctx.verify_spend_batch(sk, proofs_host[:PROOF_BYTES * 8], 1); t1 = (time.perf_counter() - t) / 8
    assert st == bytes(8)
    n = max(cores, min(len(proofs_host) // PROOF_BYTES, int(seconds_target / t1 * cores * 0.6)))
    n = min(n, len(proofs_host) // PROOF_BYTES)
    t = time.perf_counter(); st = ctx.verify_spend_batch(sk, proofs_host[:PROOF_BYTES * n], cores); dt = time.perf_counter() - t
    assert st == bytes(n)
    try:
        os.unlink(native)
    except OSError:
        pass
    return {"value": n / dt, "unit": "verifies/s", "cores": cores, "kind": "port",
            "sample": "%d of the bench's own L=128 proofs, C oracle (-O3 -march=native), %d threads, %.1f s; 1 thread: %.2f verifies/s"
                      % (n, cores, dt, 1.0 / t1),
            "single_thread_value": 1.0 / t1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--distinct", type=int, default=4096)
    ap.add_argument("--max-batch", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # test hooks: exercise the N>1 control path on a box with one GPU (RCCL refuses two ranks on one device)
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--force-device", type=int, default=-1)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")
    if args.force_device >= 0:
        local = args.force_device
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    from act_amd import capi
    n = 1 << args.batch_log2
    h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01", device=local)    # benches/benchmark.rs:9-16
    eng = capi.Engine(h, L, device=local, max_batch=args.max_batch, transcript=capi.TRANSCRIPT_DEVICE)
    sk = eng.private_key_random(shake("bench-sk", 64))
    distinct = min(args.distinct, n)
    proofs = make_inputs(eng, sk, distinct)
    host = np.frombuffer(proofs, np.uint8).reshape(distinct, PROOF_BYTES)
    dev = torch.from_numpy(host.copy()).cuda().repeat(n // distinct, 1).contiguous()      # distinct proofs tiled
    # 1 lane in 1024 tampered: flipped charge bit (-> InvalidClientSpendProof) or A' = identity (-> IdentityPointError)
    idx = torch.arange(513, n, 1024, device="cuda")
    dev[idx[0::2], 32] ^= 1
    dev[idx[1::2], 64:96] = 0
    expect = torch.zeros(n, dtype=torch.uint8, device="cuda"); expect[idx[0::2]] = 7; expect[idx[1::2]] = 6
    status = torch.zeros(n, dtype=torch.uint8, device="cuda")

    def step():
        eng.verify_spend_dev(sk, n, dev.data_ptr(), status.data_ptr())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()      # the engine runs on its own streams: inputs written by torch must be complete first
    for _ in range(args.warmup):
        step()
    eng.prof_reset(); eng.prof_enable(True)        # HIP events on the engine's own stream (torch events cannot see it)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.equal(status, expect), "verification statuses wrong"
    prof = eng.prof()

    if rank == 0:
        value = world * n * args.steps / elapsed
        bits = prof.get("k_spend_bits", {"ms": 0.0, "launches": 1, "lanes": 0})
        launch_s = bits["ms"] / 1e3 / max(1, bits["launches"])
        proofs_per_launch = bits["lanes"] / max(1, bits["launches"]) / L
        achieved = ALGO_BYTES_PER_VERIFY * proofs_per_launch / launch_s / 1e9 if launch_s else 0.0
        kernel_ms = {k: round(v["ms"] / args.steps, 3) for k, v in prof.items()}
        # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE);
        # PMC counters cannot be collected inside this process, so the figure is read from profiles/ when its launch size matches
        traffic = None
        valu = None
        try:
            v = json.load(open(os.path.join(ROOT, "profiles", "r01_h_pmc_valu.json")))
            # the roofline that actually binds: SIMD issue slots.  A wavefront of k_spend_bits is v instructions at c cycles
            # each with two waves per SIMD; at the measured clock the chip's 1024 SIMDs cannot exceed this many verifies/s.
            per_wave, cyc, clk = v["valu_instructions_per_wave"], v["cycles_per_valu_instruction_per_simd_2waves"], v["effective_clock_ghz"] * 1e9
            waves_per_proof = L / 64.0
            bound = 1024 * clk / (waves_per_proof * per_wave * cyc)
            valu = {"valu_instructions_per_wave": per_wave, "cycles_per_instruction": cyc, "clock_ghz": v["effective_clock_ghz"],
                    "issue_bound_verifies_per_s_per_gpu": bound, "frac_of_issue_bound": (value / world) / bound,
                    "source": "profiles/r01_h_pmc_valu.json (rocprofv3 --pmc SQ_INSTS_VALU, SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE)"}
        except (OSError, KeyError, ValueError):
            pass
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "r01_h_pmc_hbm_traffic.json")))
            if int(t["proofs_per_launch"]) == int(proofs_per_launch):
                traffic = t["hbm_bytes_per_launch_fetch_x2"]
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "spend-proof verifies/sec (whole node), batch=2^20", "value": value, "unit": "verifies/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs / u64 accumulators (integer)",
            "data": "synthetic: %d distinct valid L=128 proofs made by the engine's own prover, tiled to 2^%d per GPU, 1/1024 lanes tampered; device transcripts"
                    % (distinct, args.batch_log2),
            "config": {"workload": "configs[1] scaled to the metric batch: 2^%d spend-proof verifies per GPU, L=128 (the crate's width), inputs resident in HBM"
                                   % args.batch_log2,
                       "batch_per_gpu": n, "range_bits": L, "lanes_per_launch": args.max_batch, "transcript": "device BLAKE3", "sharding": "independent batches per rank, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": "k_spend_bits", "avg_launch_ms": 1e3 * launch_s, "proofs_per_launch": proofs_per_launch,
                         "algorithmic_bytes_per_verify": ALGO_BYTES_PER_VERIFY,
                         "concurrent_launches": 2, "valu_issue": valu,
                         "note": "integer-VALU-issue bound, not HBM bound (DESIGN.md 6): ~0.78 M VALU instructions per wavefront-lane; two chunks' launches overlap on two streams, so avg_launch_ms is per overlapped launch; traffic (bytes, PMC) is dominated by the per-lane Pippenger buckets cycling through L2/Infinity Cache"},
            "kernel_ms_per_step": kernel_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(proofs, h, sk)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
