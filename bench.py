#!/usr/bin/env python3
"""bench.py — spend-proof verifies/sec (whole node), batch 2^20 per GPU, L = 128 (BASELINE.json metric).

A step = one pass of the hot path (PrivateKey::refund up to the challenge check, src/lib.rs:787-844) over one batch of
2^20 synthetic spend proofs that are already resident in HBM when the timed region starts, transcripts hashed by the
device BLAKE3 kernel (byte-identical to the host path).  That is `value`.  One process per GPU; ranks shard independent
batches (weak scaling, no data-path collective — torch.distributed/RCCL is used only for the barrier and the
max-over-ranks reduction of the timing).

At N = 1 the same process also measures and prints, inside the one JSON line:
  extra.host_transcript_hbm      the library's default / north-star contract mode: transcripts hashed on host threads
  extra.host_transcript_hostmem  ... with the proofs in pinned HOST memory and statuses returned to host memory
                                 (ACT_MEM_HOST, what a Rust caller passes): the PCIe-inclusive rate, never `value`
  extra.refund                   verify + BBS re-sign (src/lib.rs:787-868), device transcripts, HBM-resident
  extra.verify_L64               BASELINE config 2: 2^16 verifies at L = 64
  roofline                       HBM view the contract asks for (algorithmic bytes / k_spend_bits busy time) ...
  roofline.alu                   ... and the roofline that actually binds: 64-bit multiply-accumulates per second against
                                 a v_mad_u64_u32 micro-kernel timed in this run on this GPU; the multiply-accumulates
                                 per verify are counted, not estimated (the kernels' own lane bodies executed on the
                                 host with counting field operations, tests/hostcheck)
  cpu_baseline                   the C oracle (a port of the reference algorithm) on this box's host cores

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import ctypes
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ELL = 2**252 + 27742317777372353535851937790883648493
CSRC = os.path.join(ROOT, "anonymous-credit-tokens_amd", "csrc")
# sources that determine k_spend_bits: PMC summaries under profiles/ are only cited when they were taken from these bytes
KERNEL_SOURCES = ["fe25519.h", "fe25519_gen.inc", "fe25519_consts.inc", "sc25519.h", "ge25519.h", "msm.h", "kernels.h", "spend_lanes.h",
                  "k_spend_verify.hip"]


def proof_bytes(L):
    return 32 * (14 + 4 * L)          # algorithmic bytes in per verify (+ 1 status byte out)


def shake(label, n):
    return hashlib.shake_256(label.encode()).digest(n)


def scb(v):
    return (v % ELL).to_bytes(32, "little")


def kernel_source_sha16():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


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


def cpu_baseline(proofs_host, h, sk, L, seconds_target=12.0):
    """The C oracle (a restatement of the reference algorithm with the reference's operation structure — NOT the
    Rust crate, which cannot be built here) timed on this box's host cores on a bounded sample of the same proofs."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    pb = proof_bytes(L)
    native = os.path.join("/tmp", "libact_oracle_native_%d.so" % os.getpid())
    try:
        oracle_c.build(native_out=native)
        o = oracle_c.Oracle(native)
    except Exception:
        o = oracle_c.Oracle()
    ctx = o.ctx(h, L)
    cores = usable_cores()
    t = time.perf_counter(); st = ctx.verify_spend_batch(sk, proofs_host[:pb * 8], 1); t1 = (time.perf_counter() - t) / 8
    assert st == bytes(8)
    n = max(cores, min(len(proofs_host) // pb, int(seconds_target / t1 * cores * 0.6)))
    n = min(n, len(proofs_host) // pb)
    t = time.perf_counter(); st = ctx.verify_spend_batch(sk, proofs_host[:pb * n], cores); dt = time.perf_counter() - t
    assert st == bytes(n)
    per_fn = config1_round_trip(ctx, L)
    try:
        os.unlink(native)
    except OSError:
        pass
    return {"value": n / dt, "unit": "verifies/s", "cores": cores, "kind": "port", "config1_single_round_trip_ms": per_fn,
            "sample": "%d of the bench's own L=%d proofs, C oracle (-O3 -march=native), %d threads, %.1f s; 1 thread: %.2f verifies/s"
                      % (n, L, cores, dt, 1.0 / t1),
            "single_thread_value": 1.0 / t1}


def config1_round_trip(ctx, L, reps=6):
    """BASELINE configs[0]: the single issue -> prove_spend -> refund round trip of benches/benchmark.rs:34-212 on ONE host
    thread, each function timed on its own (ms per call, C oracle = port of the reference algorithm): bench Params, credit
    in [20, 1000), charge in [1, c-1] (benches/benchmark.rs:131, 147-154)."""
    import random
    r = random.Random(1)
    acc = {k: 0.0 for k in ("request", "issue", "issuance_to_credit_token", "prove_spend", "refund", "refund_to_credit_token")}

    def t(key, fn):
        t0 = time.perf_counter(); out = fn(); acc[key] += time.perf_counter() - t0
        return out
    sk = ctx.private_key_random(shake("c1-sk", 64))
    for i in range(reps):
        c = r.randrange(20, 1000); s = r.randrange(1, c)
        pre = ctx.pre_issuance_random(shake("c1-pre%d" % i, 128))
        req = t("request", lambda: ctx.request(pre, shake("c1-rq%d" % i, 128)))
        st, resp = t("issue", lambda: ctx.issue(sk, req, scb(c), shake("c1-ir%d" % i, 128)))
        st, tok = t("issuance_to_credit_token", lambda: ctx.issuance_to_credit_token(pre, sk[32:], req, resp))
        rng = shake("c1-pr%d" % i, 64 * (4 * L + 12))
        st, proof, prer = t("prove_spend", lambda: ctx.prove_spend(tok, scb(s), rng))
        st, rf = t("refund", lambda: ctx.refund(sk, proof, shake("c1-rr%d" % i, 128)))
        assert st == 0
        st, tok2 = t("refund_to_credit_token", lambda: ctx.refund_to_credit_token(prer, proof, rf, sk[32:]))
        assert st == 0
    return {k: round(1e3 * v / reps, 3) for k, v in acc.items()}


def count_field_ops(h, L, sk, proofs_host, fb_bits, sample=4):
    """Exact field-operation counts of one verify: the spend kernels' own lane bodies (csrc/spend_lanes.h) executed on the
    host, with counting fe_mul / fe_sq, by the instrumented test build tests/hostcheck (built here with g++).  A count of
    operations, not a computation of results: statuses come from the GPU.  fb_bits = the context's table window widths
    (g, h1, h2, h3): a fixed-base product is ceil(253 / bits) mixed additions of 7 multiplications."""
    src = os.path.join(ROOT, "tests", "hostcheck", "hostcheck.cpp")
    out = os.path.join("/tmp", "libhostcheck_bench_%d.so" % os.getpid())
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", out, src], check=True)
    hc = ctypes.CDLL(out)
    pb = proof_bytes(L); n = sample
    tb = 184 + 40 * (6 + 3 * L)
    tr = ctypes.create_string_buffer(n * tb); st = ctypes.create_string_buffer(n); kp = ctypes.create_string_buffer(32 * n)
    c = (ctypes.c_uint64 * 25)()
    ok = hc.hc_spend_verify(h, L, sk, n, proofs_host[:pb * n], tr, st, kp, c)
    os.unlink(out)
    assert ok == 1 and st.raw == bytes(n)
    windows = [-(-253 // b) for b in fb_bits]
    per = {}
    for k, name in enumerate(("k_spend_prep", "k_spend_bits", "k_spend_enc", "k_spend_tail")):
        mul, sq = c[6 * k], c[6 * k + 1]
        fb = [c[6 * k + 2 + b] for b in range(4)]
        per[name] = {"fe_mul": (mul - sum(fb[b] * (c[24] - windows[b]) * 7 for b in range(4))) / n, "fe_sq": sq / n}
    return per


def newest_matching_pmc(kind, proofs_per_launch, sha, L=128):
    """Newest profiles/*_<kind>.json collected from the kernel sources this run was built from (same hash, same launch size,
    same range width)."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s.json" % kind))):
        try:
            j = json.load(open(f))
        except (OSError, ValueError):
            continue
        if j.get("kernel_source_sha16") == sha and int(j.get("proofs_per_launch", -1)) == int(proofs_per_launch) and int(j.get("range_bits", 128)) == L:
            best = (os.path.relpath(f, ROOT), j)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--range-bits", type=int, default=128, help="L; 128 is the crate's width and the metric's")
    ap.add_argument("--distinct", type=int, default=4096)
    ap.add_argument("--max-batch", type=int, default=65536)
    ap.add_argument("--extra-log2", type=int, default=18, help="proofs per extra measurement (contract mode, refund)")
    ap.add_argument("--pipeline-depth", type=int, default=2, help="chunks in flight; 1 for profiling runs (rocprofv3 per-kernel durations then do not overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    # test hooks: exercise the N>1 control path on a box with one GPU (RCCL refuses two ranks on one device)
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--force-device", type=int, default=-1)
    args = ap.parse_args()
    L = args.range_bits
    PB = proof_bytes(L)

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
    eng.set_pipeline_depth(args.pipeline_depth)
    sk = eng.private_key_random(shake("bench-sk", 64))
    distinct = min(args.distinct, n)
    proofs = make_inputs(eng, sk, distinct)
    host = np.frombuffer(proofs, np.uint8).reshape(distinct, PB)
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
    eng.prof_reset(); eng.prof_enable(True)        # HIP events on the engine's own streams (torch events cannot see them)
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
        ms_per_step = 1e3 * elapsed / args.steps
        bits = prof.get("k_spend_bits", {"ms": 0.0, "busy_ms": 0.0, "launches": 1, "lanes": 0})
        launches_per_step = bits["launches"] / args.steps
        # launches of the two chunks in flight overlap, so a launch's own event-to-event duration double counts; the time
        # during which the kernel was executing at all (union of the launch intervals), divided by the launches, does not
        launch_s = bits["busy_ms"] / 1e3 / max(1, bits["launches"])
        proofs_per_launch = bits["lanes"] / max(1, bits["launches"]) / L
        algo_bytes = PB + 1
        achieved = algo_bytes * proofs_per_launch / launch_s / 1e9 if launch_s else 0.0
        kernel_ms = {k: {"busy": round(v["busy_ms"] / args.steps, 3), "sum_of_launches": round(v["ms"] / args.steps, 3)} for k, v in prof.items()}
        assert bits["busy_ms"] / args.steps <= ms_per_step * 1.001, "kernel busy time exceeds the step time"

        out = {
            "metric": "spend-proof verifies/sec (whole node), batch=2^20", "value": value, "unit": "verifies/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs / u64 accumulators (integer)",
            "data": "synthetic: %d distinct valid L=%d proofs made by the engine's own prover, tiled to 2^%d per GPU, 1/1024 lanes tampered; device transcripts"
                    % (distinct, L, args.batch_log2),
            "config": {"workload": "configs[1] scaled to the metric batch: 2^%d spend-proof verifies per GPU, L=%d%s, inputs resident in HBM"
                                   % (args.batch_log2, L, " (the crate's width)" if L == 128 else ""),
                       "batch_per_gpu": n, "range_bits": L, "lanes_per_launch": args.max_batch, "chunks_in_flight": args.pipeline_depth, "transcript": "device BLAKE3",
                       "fixed_base_window_bits_g_h1_h2_h3": eng.fixed_base_bits(),
                       "sharding": "independent batches per rank, no collective"},
        }
        roof = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                "kernel": "k_spend_bits", "avg_launch_ms": 1e3 * launch_s, "launches_per_step": launches_per_step,
                "avg_launch_ms_x_launches_per_step": 1e3 * launch_s * launches_per_step,
                "proofs_per_launch": proofs_per_launch, "algorithmic_bytes_per_verify": algo_bytes,
                "timing": "HIP events on the engine's streams over the timed region; avg_launch_ms = (time during which k_spend_bits was executing) / launches "
                          "— two chunks' launches overlap on two streams, each launch's own start-to-end duration is kernel_ms_per_step.sum_of_launches",
                "note": "not HBM bound: 16.8 KB in per verify against ~43 M 64-bit multiply-accumulates (roofline.alu is the binding view); "
                        "PMC traffic is dominated by the per-lane Pippenger buckets cycling through L2 / Infinity Cache"}
        sha = kernel_source_sha16()
        t = newest_matching_pmc("pmc_hbm_traffic", proofs_per_launch, sha, L)
        if t:
            roof["traffic"] = t[1]["hbm_bytes_per_launch_fetch_x2"]; roof["traffic_source"] = t[0]
        else:
            roof["traffic_source"] = "none: no profiles/*_pmc_hbm_traffic.json was collected from these kernel sources (sha %s) at L = %d" % (sha, L)
        v = newest_matching_pmc("pmc_valu", proofs_per_launch, sha, L)
        if v:
            j = v[1]
            roof["pmc_valu"] = {"source": v[0], "valu_instructions_per_wave": j["valu_instructions_per_wave"],
                                "cycles_per_valu_instruction_per_simd_2waves": j["cycles_per_valu_instruction_per_simd_2waves"],
                                "effective_clock_ghz": j["effective_clock_ghz"], "launch_ms_solo_under_pmc": j["launch_ms_under_pmc"].get("SQ_INSTS_VALU")}
        out["roofline"] = roof
        out["kernel_ms_per_step"] = kernel_ms
        out["kernel_source_sha16"] = sha

        if world == 1:
            # ---- the ALU roofline: measured peak of the multiply-accumulate instruction, counted work per verify -------
            peak_mad, probe_ms = capi.ubench_mad(local)
            ops = count_field_ops(h, L, sk, proofs, eng.fixed_base_bits())
            fe_mul = sum(v["fe_mul"] for v in ops.values()); fe_sq = sum(v["fe_sq"] for v in ops.values())
            mad_per_verify = 100 * fe_mul + 55 * fe_sq          # fe25519.h: a product is 10 columns x 10 v_mad_u64_u32, a square 55
            bits_mad = (100 * ops["k_spend_bits"]["fe_mul"] + 55 * ops["k_spend_bits"]["fe_sq"]) * proofs_per_launch
            roof["alu"] = {"fe_mul_per_verify": fe_mul, "fe_sq_per_verify": fe_sq, "mad_per_verify": mad_per_verify,
                           "achieved_mad_per_s": value * mad_per_verify, "peak_mad_per_s": peak_mad, "frac": value * mad_per_verify / peak_mad,
                           "k_spend_bits_frac": bits_mad / launch_s / peak_mad if launch_s else None,
                           "unit": "lane multiply-accumulates (v_mad_u64_u32) per second", "probe_ms": probe_ms,
                           "per_kernel_field_ops_per_verify": ops,
                           "how": "peak: act_ubench_mad_u64_u32 (8 register-resident accumulators per lane advanced by blocks of 10 dependent multiply-accumulates, 8 waves per SIMD, ~0.3 s so that the clock settles) timed in this process; "
                                  "work: fe_mul / fe_sq executed by the kernels' own lane bodies, counted on the host (tests/hostcheck), x 100 / 55 "
                                  "multiply-accumulates each; k_spend_bits_frac uses that kernel's busy time alone"}

        if world == 1 and not args.no_extras:
            out["extra"] = extras(args, eng, capi, torch, np, sk, dev, expect, h, local, L, PB)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(proofs, h, sk, L)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def timed(fn, sync):
    fn(); sync()                       # warm-up (buffers grow, pinned staging is allocated)
    t = time.perf_counter(); fn(); sync()
    return time.perf_counter() - t


def extras(args, eng, capi, torch, np, sk, dev, expect, h, local, L, PB):
    """Rates the north-star contract and SURVEY.md 8d ask for beside the headline, each over 2^extra_log2 proofs, N = 1 only."""
    m = min(1 << args.extra_log2, dev.shape[0])
    sync = torch.cuda.synchronize
    ex = {"proofs_each": m}
    st = torch.zeros(m, dtype=torch.uint8, device="cuda")
    # (1) host transcripts (library default, src/transcript.rs stays on the host), inputs in HBM
    eng.set_transcript_mode(capi.TRANSCRIPT_HOST)
    dt = timed(lambda: eng.verify_spend_dev(sk, m, dev.data_ptr(), st.data_ptr()), sync)
    assert torch.equal(st, expect[:m])
    ex["host_transcript_hbm"] = {"value": m / dt, "unit": "verifies/s", "what": "ACT_TRANSCRIPT_HOST, proofs and statuses in HBM (ACT_MEM_DEVICE)"}
    nfull = dev.shape[0]
    if nfull > m:                         # the same at the metric batch: pipeline fill and drain amortised over 16 chunks
        stf = torch.zeros(nfull, dtype=torch.uint8, device="cuda")
        dt = timed(lambda: eng.verify_spend_dev(sk, nfull, dev.data_ptr(), stf.data_ptr()), sync)
        assert torch.equal(stf, expect)
        ex["host_transcript_hbm_metric_batch"] = {"value": nfull / dt, "unit": "verifies/s", "proofs": nfull,
                                                  "what": "ACT_TRANSCRIPT_HOST over the whole 2^%d batch, proofs in HBM" % args.batch_log2}
    # (2) ... with proofs in pinned host memory and statuses back in host memory: what a Rust caller's slices are
    hp = torch.empty((m, PB), dtype=torch.uint8, pin_memory=True); hp.copy_(dev[:m]); sync()
    hs = torch.zeros(m, dtype=torch.uint8, pin_memory=True)
    dt = timed(lambda: eng.verify_spend_ptr(sk, m, capi.MEM_HOST, hp.data_ptr(), hs.data_ptr()), sync)
    assert torch.equal(hs, expect[:m].cpu())
    ex["host_transcript_hostmem"] = {"value": m / dt, "unit": "verifies/s", "pcie_GBps": m * PB / dt / 1e9, "host_threads": usable_cores(),
                                     "what": "ACT_TRANSCRIPT_HOST + ACT_MEM_HOST (pinned): the contract mode end to end, PCIe and host BLAKE3 inclusive"}
    eng.set_transcript_mode(capi.TRANSCRIPT_DEVICE)
    dt = timed(lambda: eng.verify_spend_ptr(sk, m, capi.MEM_HOST, hp.data_ptr(), hs.data_ptr()), sync)
    assert torch.equal(hs, expect[:m].cpu())
    ex["device_transcript_hostmem"] = {"value": m / dt, "unit": "verifies/s", "pcie_GBps": m * PB / dt / 1e9,
                                       "what": "ACT_TRANSCRIPT_DEVICE + ACT_MEM_HOST (pinned)"}
    del hp
    # (3) refund = verify + sign (src/lib.rs:787-868), per-lane rng resident in HBM
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    rng = torch.randint(0, 256, (m, 128), dtype=torch.uint8, device="cuda", generator=g)
    rf = torch.zeros((m, 128), dtype=torch.uint8, device="cuda")
    sync()
    dt = timed(lambda: eng.refund_dev(sk, m, dev.data_ptr(), rng.data_ptr(), capi.RNG_PER_LANE, rf.data_ptr(), st.data_ptr()), sync)
    assert torch.equal(st, expect[:m])
    assert bool((rf[expect[:m] != 0] == 0).all()) and bool((rf[expect[:m] == 0].any(dim=1)).all())
    ex["refund"] = {"value": m / dt, "unit": "refunds/s", "what": "verify + BBS re-sign, device transcripts, HBM-resident, ACT_RNG_PER_LANE"}
    # (4) BASELINE config 2: 2^16 verifies at L = 64
    if L == 128:
        e64 = capi.Engine(h, 64, device=local, max_batch=args.max_batch, transcript=capi.TRANSCRIPT_DEVICE)
        d64 = 1024
        p64 = make_inputs(e64, sk, d64)
        n64 = 1 << 16
        dev64 = torch.from_numpy(np.frombuffer(p64, np.uint8).reshape(d64, proof_bytes(64)).copy()).cuda().repeat(n64 // d64, 1).contiguous()
        st64 = torch.zeros(n64, dtype=torch.uint8, device="cuda")
        sync()
        dt = timed(lambda: e64.verify_spend_dev(sk, n64, dev64.data_ptr(), st64.data_ptr()), sync)
        assert int(st64.sum()) == 0
        ex["verify_L64"] = {"value": n64 / dt, "unit": "verifies/s", "what": "BASELINE configs[1]: 2^16 spend-proof verifies, 64-bit range, one launch chunk, device transcripts"}
        e64.close()
    return ex


if __name__ == "__main__":
    main()
