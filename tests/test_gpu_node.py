"""Node-level dispatch (include/act_mi355x.h act_node_*): a batch cut into contiguous shards over several contexts must
give byte-for-byte what one context gives -- in particular with ACT_RNG_SEQUENTIAL, where a lane's rng slice depends on
how many lanes in front of it (on other shards too) were accepted (/root/reference/src/lib.rs:638-643, 842-846).  A box
of the test pool has one GPU, so the "devices" are several contexts on device 0: the dispatcher, its threads, the slicing
and the two-phase sequential path are exactly those of a multi-GPU node; and bench.py's N > 1 path is driven with two
ranks sharing device 0 through its gloo hook."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, shake, scb

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [(0, 0), (0, 0, 0)])
@pytest.mark.parametrize("mode", [0, 1])
def test_node_equals_single_context(engine_factory, bench_params, devices, mode):
    from act_amd import capi
    L, N = 8, 23
    eng = engine_factory(bench_params, L, max_batch=5, transcript=mode)
    node = capi.Node(bench_params, L, devices=devices, max_batch=5, transcript=mode)
    assert node.device_count() == len(devices)
    sk = eng.private_key_random(shake("node-sk", 64))
    pre = eng.pre_issuance_random(shake("node-pre", 128 * N))
    rq = shake("node-rq", 128 * N)
    req = eng.request(pre, rq)
    assert node.request(pre, rq) == req
    bad = bytearray(req)
    bad[128 * 1 + 70] ^= 1; bad[128 * 9 + 40] ^= 1; bad[128 * 10 + 3] ^= 0x08; bad[128 * 22 + 100] ^= 1      # rejected lanes on every shard
    bad = bytes(bad)
    cam = b"".join(scb(200 + i) for i in range(N))
    irng = shake("node-ir", 128 * N)
    for rng_mode in (capi.RNG_PER_LANE, capi.RNG_SEQUENTIAL):
        want = eng.issue(sk, bad, cam, irng, rng_mode)
        assert node.issue(sk, bad, cam, irng, rng_mode) == want, rng_mode
        assert {0, 1} <= set(want[0])
    st, resp = eng.issue(sk, req, cam, irng)
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    assert node.issuance_to_credit_token(pre, sk[32:], req, resp) == (st, tok)
    s_b = b"".join(scb(3 * i) for i in range(N))
    prng = shake("node-pr", eng.prove_rng_bytes * N)
    want = eng.prove_spend(tok, s_b, prng)
    assert node.prove_spend(tok, s_b, prng) == want
    proofs, prer = bytearray(want[1]), want[2]
    pb = eng.proof_bytes
    proofs[pb * 0 + 33] ^= 2; proofs[pb * 7 + 64:pb * 7 + 96] = bytes(32); proofs[pb * 8 + 32 * 5 + 1] ^= 0x40
    proofs[pb * 15 + 32 * (4 + L) + 2] ^= 1; proofs[pb * 22 + 33] ^= 1
    proofs = bytes(proofs)
    assert node.verify_spend(sk, proofs, True) == eng.verify_spend(sk, proofs, True)
    rrng = shake("node-rr", 128 * N)
    for rng_mode in (capi.RNG_PER_LANE, capi.RNG_SEQUENTIAL):
        want = eng.refund(sk, proofs, rrng, rng_mode)
        got = node.refund(sk, proofs, rrng, rng_mode)
        assert got == want, rng_mode
        assert {0, 6, 7} <= set(want[0])
    # the halves through the node handle: check, draw exactly 128 bytes per accepted lane, sign -- what the Rust binding does
    st, kp = node.verify_spend(sk, proofs, True)
    accepted = sum(1 for v in st if v == 0)
    assert node.refund_sign(sk, kp, st, rrng[:128 * accepted]) == eng.refund(sk, proofs, rrng, capi.RNG_SEQUENTIAL)
    st = node.issue_check(bad)
    accepted = sum(1 for v in st if v == 0)
    assert node.issue_sign(sk, bad, cam, st, irng[:128 * accepted]) == eng.issue(sk, bad, cam, irng, capi.RNG_SEQUENTIAL)
    st, rf = eng.refund(sk, proofs, rrng)
    assert node.refund_to_credit_token(prer, proofs, rf, sk[32:]) == eng.refund_to_credit_token(prer, proofs, rf, sk[32:])
    # fewer lanes than shards, and an empty batch
    assert node.refund(sk, proofs[:pb], rrng[:128], capi.RNG_SEQUENTIAL) == eng.refund(sk, proofs[:pb], rrng[:128], capi.RNG_SEQUENTIAL)
    assert node.verify_spend(sk, b"") == b""
    node.close()


def test_check_then_sign_equals_one_call(engine_factory, bench_params):
    """act_issue_check_batch + act_issue_sign_batch / act_verify_spend_batch + act_refund_sign_batch = the one-call forms."""
    import ctypes as C
    import numpy as np
    from act_amd import capi
    L, N = 8, 11
    eng = engine_factory(bench_params, L, max_batch=4)
    lib = eng.lib
    sk = eng.private_key_random(shake("cs-sk", 64))
    pre = eng.pre_issuance_random(shake("cs-pre", 128 * N)); req = bytearray(eng.request(pre, shake("cs-rq", 128 * N)))
    req[128 * 2 + 70] ^= 1; req = bytes(req)
    cam = scb(77) * N; irng = shake("cs-ir", 128 * N)
    buf = lambda b: np.frombuffer(b, np.uint8).copy()
    for rng_mode in (0, 1):
        want = eng.issue(sk, req, cam, irng, rng_mode)
        st = np.zeros(N, np.uint8); out = np.zeros(160 * N, np.uint8); st2 = np.zeros(N, np.uint8)
        a = [buf(x) for x in (sk, req, cam, irng)]
        assert lib.act_issue_check_batch(eng.ctx, N, 0, a[1].ctypes.data, st.ctypes.data) == 0
        assert lib.act_issue_sign_batch(eng.ctx, N, 0, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, st.ctypes.data, a[3].ctypes.data, rng_mode,
                                        out.ctypes.data, st2.ctypes.data) == 0
        assert (st2.tobytes(), out.tobytes()) == want
    st, resp = eng.issue(sk, eng.request(pre, shake("cs-rq", 128 * N)), cam, irng)
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], eng.request(pre, shake("cs-rq", 128 * N)), resp)
    st, proofs, _ = eng.prove_spend(tok, scb(5) * N, shake("cs-pr", eng.prove_rng_bytes * N))
    proofs = bytearray(proofs); proofs[eng.proof_bytes * 4 + 33] ^= 1; proofs = bytes(proofs)
    rrng = shake("cs-rr", 128 * N)
    for rng_mode in (0, 1):
        want = eng.refund(sk, proofs, rrng, rng_mode)
        st, kp = eng.verify_spend(sk, proofs, True)
        a = [buf(x) for x in (sk, kp, st, rrng)]
        out = np.zeros(128 * N, np.uint8); st2 = np.zeros(N, np.uint8)
        assert lib.act_refund_sign_batch(eng.ctx, N, 0, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, rng_mode,
                                         out.ctypes.data, st2.ctypes.data) == 0
        assert (st2.tobytes(), out.tobytes()) == want


def test_bench_two_ranks_on_one_device():
    """bench.py's N > 1 control path (rendezvous, barrier, max-over-ranks timing, whole-job value) with two gloo ranks that
    both drive device 0 through the bench's own --dist-backend / --force-device hooks."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29671",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch-log2", "12", "--max-batch", "1024", "--distinct", "256",
           "--dist-backend", "gloo", "--force-device", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["batch_per_gpu"] == 4096
    assert abs(d["value"] - 2 * 4096 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]      # whole-job aggregate over both ranks
    assert d["roofline"]["avg_launch_ms"] * d["roofline"]["launches_per_step"] <= d["ms_per_step"] * 1.001
    assert d["roofline"]["bound"] == "valu-int-mad" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["hbm"]["bound"] == "hbm"
    # the metric as BASELINE.json words it: ONE batch over the whole node (strong scaling), 4096 / 2 proofs per rank
    s = d["strong"]
    assert s["batch_total"] == 4096 and s["batch_per_gpu"] == 2048 and s["scaling"] == "strong"
    assert abs(s["value"] - 4096 * 2 / (s["ms_per_step"] * 2 / 1e3)) < 1e-6 * s["value"]
    # --scaling strong swaps which of the two is `value`
    r = subprocess.run(cmd + ["--scaling", "strong"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "strong" and d["config"]["batch_per_gpu"] == 2048 and d["weak"]["batch_per_gpu"] == 4096
    assert d["value"] == d["strong"]["value"]


def test_bench_rccl_path_with_one_rank():
    """bench.py's RCCL calls (init_process_group("nccl", device_id=...), barrier, all_reduce MAX of the timing, destroy) on the one GPU a
    test box has: one rank under torch.distributed.run with --force-dist.  The two-rank test above uses gloo because RCCL refuses two
    ranks on one device; this one makes sure the nccl branch itself runs."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29673",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch-log2", "12", "--max-batch", "1024",
           "--force-dist", "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["distinct_proofs_per_gpu"] == 4096 and d["value"] > 0
