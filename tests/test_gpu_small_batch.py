"""The small-batch schedule (small_impl.inc spend_small_locked: calls of at most 8 192 proofs run the per-proof kernels next to the
range kernel on six streams -- the crate's own call shape is ONE proof per call, /root/reference/src/lib.rs:781-786) against the
pipelined schedule and the C oracle: statuses, enc(K'), refunds under both rng modes and the complete transcripts must be the same
bytes whichever schedule runs them.  Every rejection kind, ragged widths, single-proof calls."""
import pytest

from conftest import ELL, shake, scb

pytestmark = pytest.mark.gpu

MODES = [0, 1]   # ACT_TRANSCRIPT_HOST, ACT_TRANSCRIPT_DEVICE


def _tampered_batch(eng, octx, sk, L, N, tag):
    pre = eng.pre_issuance_random(shake(tag + "-pre", 128 * N)); req = eng.request(pre, shake(tag + "-rq", 128 * N))
    amounts = [(1 << min(L, 60)) // (i + 1) + i for i in range(N)]
    st, resp = eng.issue(sk, req, b"".join(scb(a) for a in amounts), shake(tag + "-ir", 128 * N))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    assert st == bytes(N)
    spend = [a // 3 for a in amounts]
    spend[4] = amounts[4] + 1                               # overspend: produced, then rejected
    st, proofs, _ = eng.prove_spend(tok, b"".join(scb(v) for v in spend), shake(tag + "-pr", eng.prove_rng_bytes * N))
    assert st == bytes(N)
    pb = eng.proof_bytes
    t = bytearray(proofs)
    t[pb * 1 + 33] ^= 2                                     # s
    t[pb * 3 + 64:pb * 3 + 96] = bytes(32)                  # A' = identity -> 6
    t[pb * 8 + 32 * (4 + (L - 1)) + 9] ^= 0x40              # last Com: almost surely undecodable -> 255
    t[pb * 10 + 32 * (12 + L) + 5] ^= 1                     # gamma0[0]
    t[pb * 12 + 96 + 1] ^= 0x20                             # B_bar
    t[pb * 13 + 32 * (4 + min(2, L - 1)):pb * 13 + 32 * (5 + min(2, L - 1))] = bytes(32)      # a Com = identity (the base the d-free additions cannot take)
    t[pb * 15 + 32 * (12 + 2 * L):pb * 15 + 32 * (13 + 2 * L)] = b"\xff" * 32                 # z[0][0] not canonical
    return bytes(t)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("L", [128, 64, 8, 100, 3])
def test_both_schedules_give_the_oracle_bytes(engine_factory, oracle, bench_params, L, mode):
    octx = oracle.ctx(bench_params, L)
    N = 23
    eng = engine_factory(bench_params, L, max_batch=32, transcript=mode)            # 23 <= 32: one chunk
    sk = octx.private_key_random(shake("sb-pk-%d" % L, 64))
    t = _tampered_batch(eng, octx, sk, L, N, "sb-%d" % L)
    pb = eng.proof_bytes
    rrng = shake("sb-rr-%d" % L, 128 * N)
    want = [octx.verify_spend(sk, t[pb * i:pb * i + pb], True) for i in range(N)]
    assert {0, 6, 7} <= {w[0] for w in want}
    try:
        for small_max in (8192, 0):                          # the small-batch schedule, then the pipelined one on the same engine
            eng.set_small_batch_max(small_max)
            st, kp = eng.verify_spend(sk, t, True)
            trs = eng.last_spend_transcripts(N)
            for i in range(N):
                so, kpo, tro = want[i]
                assert st[i] == so, (small_max, i)
                assert kp[32 * i:32 * i + 32] == (kpo if so == 0 else bytes(32)), (small_max, i)
                if so in (0, 7):                              # the reference builds no transcript for an identity A' or an undecodable point
                    assert trs[i] == tro, (small_max, i)
            for rng_mode in (0, 1):
                st, rf = eng.refund(sk, t, rrng, rng_mode)
                cur = 0
                for i in range(N):
                    slot = i if rng_mode == 0 else cur
                    so, ro = octx.refund(sk, t[pb * i:pb * i + pb], rrng[128 * slot:128 * slot + 128])
                    assert so == st[i] and ro == rf[128 * i:128 * i + 128], (small_max, rng_mode, i)
                    cur += so == 0
            # the crate's call shape: one proof per call
            for i in (0, 1, 3, 8):
                st1, kp1 = eng.verify_spend(sk, t[pb * i:pb * i + pb], True)
                assert st1[0] == want[i][0] and kp1 == (want[i][1] if want[i][0] == 0 else bytes(32)), (small_max, i)
                st1, rf1 = eng.refund(sk, t[pb * i:pb * i + pb], rrng[:128])
                assert (st1[0], rf1) == octx.refund(sk, t[pb * i:pb * i + pb], rrng[:128]), (small_max, i)
        assert eng.secret_residue() == 0                      # the roles' partial sums and buckets are wiped like everything else
    finally:
        eng.set_small_batch_max(8192)


def test_small_schedule_from_device_memory_and_at_its_size_limit(engine_factory, bench_params):
    """Device-memory pointers, n = the schedule's limit and one past it (the pipelined schedule takes over): statuses agree with each
    other and with the tampering, whichever schedule ran."""
    import numpy as np
    import torch
    from act_amd import capi
    L, D = 8, 64
    eng = engine_factory(bench_params, L, max_batch=4096, transcript=capi.TRANSCRIPT_DEVICE)
    sk = eng.private_key_random(shake("sbd-sk", 64))
    pre = eng.pre_issuance_random(shake("sbd-pre", 128 * D)); req = eng.request(pre, shake("sbd-rq", 128 * D))
    st, resp = eng.issue(sk, req, scb(200) * D, shake("sbd-ir", 128 * D))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, proofs, _ = eng.prove_spend(tok, b"".join(scb(i) for i in range(D)), shake("sbd-pr", eng.prove_rng_bytes * D))
    assert st == bytes(D)
    host = np.frombuffer(proofs, np.uint8).reshape(D, eng.proof_bytes)
    try:
        eng.set_small_batch_max(2048)
        for n in (2048, 2049, 1, 63, 64, 65):
            dev = torch.from_numpy(host.copy()).cuda().repeat((n + D - 1) // D, 1)[:n].contiguous()
            idx = torch.arange(0, n, 7, device="cuda")
            dev[idx, 32] ^= 1
            status = torch.full((n,), 99, dtype=torch.uint8, device="cuda"); kp = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            eng.verify_spend_dev(sk, n, dev.data_ptr(), status.data_ptr(), kp.data_ptr())
            exp = torch.zeros(n, dtype=torch.uint8, device="cuda"); exp[idx] = 7
            assert torch.equal(status, exp), n
            assert bool((kp[idx] == 0).all()) and bool(kp[exp == 0].any(dim=1).all())
    finally:
        eng.set_small_batch_max(8192)


_CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %(root)r)
from act_amd import capi
sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
L, N = 8, 203
h = bytes.fromhex(%(h)r)
eng = capi.Engine(h, L, max_batch=256, transcript=int(sys.argv[1]))
sk = bytes.fromhex(%(sk)r)
batch = open(%(path)r, "rb").read()
st, kp = eng.verify_spend(sk, batch, True)
st2, rf = eng.refund(sk, batch, sh("sbs-rr", 128 * N), 1)
print(json.dumps({"st": st.hex(), "kp": hashlib.sha256(kp).hexdigest(), "st2": st2.hex(), "rf": hashlib.sha256(rf).hexdigest(), "residue": eng.secret_residue()}))
"""


@pytest.mark.parametrize("mode", MODES)
def test_sub_chunked_small_schedule(engine_factory, bench_params, tmp_path, mode):
    """The schedule's sub-chunking (several sub-chunks on the six streams, ACT_SMALL_SUB: a tuning knob read once per process, so a
    child process) against this process's pipelined schedule: 203 proofs in sub-chunks of 64, the last one ragged."""
    import hashlib
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    L, N = 8, 203
    eng = engine_factory(bench_params, L, max_batch=256, transcript=mode)
    sk = eng.private_key_random(shake("sbs-sk", 64))
    pre = eng.pre_issuance_random(shake("sbs-pre", 128 * N)); req = eng.request(pre, shake("sbs-rq", 128 * N))
    st, resp = eng.issue(sk, req, scb(200) * N, shake("sbs-ir", 128 * N))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, proofs, _ = eng.prove_spend(tok, b"".join(scb(i % 150) for i in range(N)), shake("sbs-pr", eng.prove_rng_bytes * N))
    assert st == bytes(N)
    t = bytearray(proofs); pb = eng.proof_bytes
    for i in range(0, N, 9):
        t[pb * i + 33] ^= 1
    t[pb * 70 + 64:pb * 70 + 96] = bytes(32)
    t = bytes(t)
    try:
        eng.set_small_batch_max(0)
        st, kp = eng.verify_spend(sk, t, True)
        st2, rf = eng.refund(sk, t, shake("sbs-rr", 128 * N), 1)
    finally:
        eng.set_small_batch_max(8192)
    assert {0, 6, 7} <= set(st)
    path = tmp_path / "batch.bin"
    path.write_bytes(t)
    src = _CHILD % {"root": ROOT, "h": bench_params.hex(), "sk": sk.hex(), "path": str(path)}
    r = subprocess.run([sys.executable, "-c", src, str(mode)], capture_output=True, text=True, timeout=600, env=dict(os.environ, ACT_SMALL_SUB="64"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["st"] == st.hex() and d["kp"] == hashlib.sha256(kp).hexdigest()
    assert d["st2"] == st2.hex() and d["rf"] == hashlib.sha256(rf).hexdigest() and d["residue"] == 0


def test_four_concurrent_callers_do_not_collapse():
    """Four threads with a context each making one-proof calls at the same time, in a process that allows HIP eight hardware queues
    per priority class (what INTEGRATION.md recommends): without the library's limit of two small calls in flight per device the
    process has more active queues than the GPU runs side by side and a call takes 20 - 60 ms instead of ~2
    (docs/history/profiles/r04_concurrent_small_calls.txt).  With it, four callers wait their turn: a few times one caller's latency."""
    import os
    import re
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", KS="1", TS="1,4")
    env.pop("ACT_SMALL_IN_FLIGHT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "concurrent_small_calls.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ms = [float(x) for x in re.findall(r"\(([0-9.]+) ms per call\)", r.stdout)]
    assert len(ms) == 2, r.stdout
    assert ms[0] < 4.0, ms                       # one caller: ~1.7 ms
    assert ms[1] < 6 * ms[0], ms                 # four callers, two in flight: ~3 x; the collapse was 13 - 40 x


@pytest.mark.parametrize("mode", MODES)
def test_threads_sharing_a_context_get_their_own_answers(engine_factory, bench_params, oracle, mode):
    """Sixteen threads share ONE context (the Rust binding keeps the handle inside `Params`, which safe code may share) and make
    calls of 1 - 3 proofs each (verify with K', refund with per-lane rng, two different keys, some proofs tampered).  The context
    serves them one at a time; every call must return exactly what the same call returns alone: statuses, enc(K'), refunds byte for
    byte -- checked against a sequential pass."""
    import random
    import threading
    L, D = 8, 48
    eng = engine_factory(bench_params, L, max_batch=256, transcript=mode)
    sks = [eng.private_key_random(shake("co-sk%d" % j, 64)) for j in range(2)]
    proofs = []
    for j, sk in enumerate(sks):
        pre = eng.pre_issuance_random(shake("co-pre%d" % j, 128 * D)); req = eng.request(pre, shake("co-rq%d" % j, 128 * D))
        st, resp = eng.issue(sk, req, b"".join(scb(60 + i) for i in range(D)), shake("co-ir%d" % j, 128 * D))
        st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
        st, pr, _ = eng.prove_spend(tok, b"".join(scb(i % 40) for i in range(D)), shake("co-pr%d" % j, eng.prove_rng_bytes * D))
        assert st == bytes(D)
        proofs.append(pr)
    pb = eng.proof_bytes
    r = random.Random(9)
    jobs = []                                   # (key index, proofs, refund?, rng)
    for t in range(16):
        mine = []
        for c in range(12):
            j = r.randrange(2); k = r.randrange(1, 4)
            lanes = [r.randrange(D) for _ in range(k)]
            blob = bytearray(b"".join(proofs[j][pb * i:pb * (i + 1)] for i in lanes))
            if r.randrange(3) == 0:
                blob[pb * r.randrange(k) + 40] ^= 1                 # a tampered scalar: InvalidClientSpendProof on that lane only
            if r.randrange(11) == 0:
                j ^= 1                                               # the other issuer's key: every lane of the call rejected
            mine.append((j, bytes(blob), r.randrange(2) == 1, shake("co-r%d-%d" % (t, c), 128 * k)))
        jobs.append(mine)

    def run_all(out):
        def work(t):
            try:
                res = []
                for j, blob, sign, rng in jobs[t]:
                    res.append(eng.refund(sks[j], blob, rng) if sign else eng.verify_spend(sks[j], blob, True))
                out[t] = res
            except BaseException as e:
                out[t] = e
        th = [threading.Thread(target=work, args=(t,)) for t in range(16)]
        for x in th:
            x.start()
        for x in th:
            x.join()

    try:
        alone = [None] * 16
        for t in range(16):                      # one thread after the other
            alone[t] = [eng.refund(sks[j], blob, rng) if sign else eng.verify_spend(sks[j], blob, True) for j, blob, sign, rng in jobs[t]]
        merged = [None] * 16
        run_all(merged)
    finally:
        pass
    for t in range(16):
        assert not isinstance(merged[t], BaseException), merged[t]
        assert merged[t] == alone[t], t
    # and the answers are not vacuous: accepted lanes, rejected lanes and refunds all occur, and one call agrees with the oracle
    flat = [st for t in range(16) for (st, _) in alone[t]]
    assert any(0 in st for st in flat) and any(7 in st for st in flat)
    j, blob, sign, rng = next(x for x in jobs[0] if x[2])
    so, rf = oracle.ctx(bench_params, L).refund(sks[j], blob[:pb], rng[:128])
    st, out = alone[0][jobs[0].index((j, blob, sign, rng))]
    assert st[0] == so and out[:128] == rf


def test_threads_sharing_a_context_issue_their_own_answers(engine_factory, bench_params, oracle):
    """PrivateKey::issue from twelve threads that share one context (a server's issuance endpoint), calls of 1 - 3 requests each,
    served one at a time under the context's lock; every caller gets exactly what its own call returns alone --
    statuses (a tampered request is rejected on its own lane only) and IssuanceResponse records byte for byte, for both keys in play,
    per-lane rng and one-lane sequential rng."""
    import random
    import threading
    from act_amd import capi
    L, D = 8, 40
    eng = engine_factory(bench_params, L, max_batch=256, transcript=1)
    octx = oracle.ctx(bench_params, L)
    sks = [eng.private_key_random(shake("ci-sk%d" % j, 64)) for j in range(2)]
    pre = eng.pre_issuance_random(shake("ci-pre", 128 * D)); req = eng.request(pre, shake("ci-rq", 128 * D))
    r = random.Random(4)
    jobs = []
    for t in range(12):
        mine = []
        for c in range(10):
            k = r.randrange(1, 4)
            lanes = [r.randrange(D) for _ in range(k)]
            blob = bytearray(b"".join(req[128 * i:128 * i + 128] for i in lanes))
            if r.randrange(3) == 0:
                blob[128 * r.randrange(k) + 40] ^= 1                # gamma: InvalidIssuanceRequestProof on that lane only
            mode = capi.RNG_SEQUENTIAL if (k == 1 and r.randrange(2)) else capi.RNG_PER_LANE
            mine.append((r.randrange(2), bytes(blob), b"".join(scb(r.randrange(1, 200)) for _ in range(k)), shake("ci-r%d-%d" % (t, c), 128 * k), mode))
        jobs.append(mine)
    try:
        alone = [[eng.issue(sks[j], blob, cam, rng, mode) for j, blob, cam, rng, mode in jobs[t]] for t in range(12)]
        j, blob, cam, rng, mode = jobs[0][0]                        # (and the oracle on one of them)
        so, ro = octx.issue(sks[j], blob[:128], cam[:32], rng[:128])
        assert (alone[0][0][0][:1], alone[0][0][1][:160]) == (bytes([so]), ro)
        merged = [None] * 12

        def work(t):
            try:
                merged[t] = [eng.issue(sks[j], blob, cam, rng, mode) for j, blob, cam, rng, mode in jobs[t]]
            except BaseException as e:
                merged[t] = e
        th = [threading.Thread(target=work, args=(t,)) for t in range(12)]
        for x in th:
            x.start()
        for x in th:
            x.join()
    finally:
        pass
    for t in range(12):
        assert not isinstance(merged[t], BaseException), merged[t]
        assert merged[t] == alone[t], t
    assert any(1 in st for res in alone for st, _ in res) and any(0 in st for res in alone for st, _ in res)
    assert eng.secret_residue() == 0
