"""Randomised soak: thousands of engine-made spend proofs with random field corruption, verified and refunded by the HIP
engine (both transcript modes) and by the C oracle; every status and every refund record must agree.  Test
infrastructure (uses oracle/); run on a GPU box:  python tools/soak.py [n] [seed]"""
import hashlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from act_amd import capi
from oracle_c import Oracle
ELL = 2**252 + 27742317777372353535851937790883648493
sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
scb = lambda v: (v % ELL).to_bytes(32, "little")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
r = random.Random(seed)
for L in [int(x) for x in os.environ.get("ACT_SOAK_L", "128,64").split(",")]:
    o = Oracle(); h = o.params_new("soak-org", "svc", "env", "v%d" % seed); octx = o.ctx(h, L)
    # ACT_SOAK_MAX_BATCH >= n: one chunk = the small-batch schedule (small_impl.inc spend_small_locked); the default, 600, pipelines chunks
    eng = capi.Engine(h, L, max_batch=int(os.environ.get("ACT_SOAK_MAX_BATCH", "600")), transcript=capi.TRANSCRIPT_DEVICE)
    sk = eng.private_key_random(sh("sk%d" % seed, 64))
    pre = eng.pre_issuance_random(sh("pre%d" % seed, 128 * n)); req = eng.request(pre, sh("rq%d" % seed, 128 * n))
    amounts = [r.randrange(1, 1 << min(L, 100)) for _ in range(n)]
    st, resp = eng.issue(sk, req, b"".join(scb(a) for a in amounts), sh("ir%d" % seed, 128 * n))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    spend = [r.randrange(0, a + 1) if r.random() < 0.9 else a + r.randrange(1, 5) for a in amounts]
    st, proofs, prer = eng.prove_spend(tok, b"".join(scb(s) for s in spend), sh("pr%d" % seed, eng.prove_rng_bytes * n))
    pb = eng.proof_bytes
    t = bytearray(proofs)
    nf = pb // 32
    for i in range(n):
        x = r.random()
        if x < 0.25:                                   # flip one bit of a random field
            f = r.randrange(nf); t[pb * i + 32 * f + r.randrange(32)] ^= 1 << r.randrange(8)
        elif x < 0.30:                                 # a field zeroed (identity point / zero scalar)
            f = r.randrange(nf); t[pb * i + 32 * f:pb * i + 32 * f + 32] = bytes(32)
        elif x < 0.33:                                 # a field from another proof
            f = r.randrange(nf); j = r.randrange(n); t[pb * i + 32 * f:pb * i + 32 * f + 32] = proofs[pb * j + 32 * f:pb * j + 32 * f + 32]
        elif x < 0.35:                                 # non-canonical bytes
            f = r.randrange(nf); t[pb * i + 32 * f:pb * i + 32 * f + 32] = bytes([255]) * 32
    t = bytes(t)
    rrng = sh("rr%d" % seed, 128 * n)
    st_o = octx.verify_spend_batch(sk, t, 16)
    ref_o = [octx.refund(sk, t[pb * i:pb * i + pb], rrng[128 * i:128 * i + 128]) for i in range(0, n, max(1, n // 256))]
    for mode in (capi.TRANSCRIPT_DEVICE, capi.TRANSCRIPT_HOST):
        eng.set_transcript_mode(mode)
        st = eng.verify_spend(sk, t)
        assert st == st_o, ("status mismatch", L, mode, [i for i in range(n) if st[i] != st_o[i]][:10])
        st2, rf = eng.refund(sk, t, rrng)
        assert st2 == st_o
        for k, i in enumerate(range(0, n, max(1, n // 256))):
            so, ro = ref_o[k]
            assert so == st2[i] and ro == rf[128 * i:128 * i + 128], ("refund mismatch", L, mode, i)
    # ACT_SOAK_THREADS=T: the same proofs once more as calls of 1 - 5 proofs from T threads that share the context (served one at a
    # time under its lock): verify with K', refund with per-lane rng, the two-call refund of the Rust binding -- every call's
    # answer against the batch answers above
    T = int(os.environ.get("ACT_SOAK_THREADS", "0"))
    if T:
        import threading
        eng.set_transcript_mode(capi.TRANSCRIPT_DEVICE)
        st_b, kp_b = eng.verify_spend(sk, t, True)
        st2_b, rf_b = eng.refund(sk, t, rrng)
        cuts, i = [], 0
        while i < n:
            k = min(n - i, r.randrange(1, 6)); cuts.append((i, k)); i += k
        errs = []
        def work(tid):
            try:
                for j, (a, k) in enumerate(cuts):
                    if j % T != tid:
                        continue
                    blob = t[pb * a:pb * (a + k)]
                    what = (a + k) % 3
                    if what == 0:
                        s1, kp = eng.verify_spend(sk, blob, True)
                        assert s1 == st_b[a:a + k] and kp == kp_b[32 * a:32 * (a + k)], ("verify", a, k)
                    elif what == 1:
                        s1, rf = eng.refund(sk, blob, rrng[128 * a:128 * (a + k)])
                        assert s1 == st2_b[a:a + k] and rf == rf_b[128 * a:128 * (a + k)], ("refund", a, k)
                    else:                                       # verify, then sign what verified (per-lane rng: the same bytes as the fused call)
                        s1, kp = eng.verify_spend(sk, blob, True)
                        out = eng.lib.act_refund_sign_batch
                        import numpy as np
                        o_rf = np.zeros(128 * k, np.uint8); o_st = np.zeros(k, np.uint8)
                        p_sk, _k1 = capi._in(sk, 64); p_kp, _k2 = capi._in(kp, 32 * k); p_st, _k3 = capi._in(s1, k); p_r, _k4 = capi._in(rrng[128 * a:128 * (a + k)], 128 * k)
                        assert out(eng.ctx, k, capi.MEM_HOST, p_sk, p_kp, p_st, p_r, capi.RNG_PER_LANE, o_rf.ctypes.data, o_st.ctypes.data) == 0
                        assert o_st.tobytes() == st2_b[a:a + k] and o_rf.tobytes() == rf_b[128 * a:128 * (a + k)], ("two-call refund", a, k)
            except BaseException as e:
                errs.append(e)
        th = [threading.Thread(target=work, args=(x,)) for x in range(T)]
        for x in th: x.start()
        for x in th: x.join()
        if errs:
            raise errs[0]
    # ACT_SOAK_TINY=1: the same proofs once more as refunds of 1 - 64 proofs -- the calls whose signature is computed beside the
    # verification (small_impl.inc spend_small_locked) -- per-lane rng from pageable memory, from pinned memory read in place, and one-proof
    # calls with the sequential convention; every answer against the batch answers above and, for one-proof calls, the oracle
    tiny_note = ""
    if os.environ.get("ACT_SOAK_TINY"):
        import numpy as np, torch
        eng.set_transcript_mode(capi.TRANSCRIPT_DEVICE)
        st2_b, rf_b = eng.refund(sk, t, rrng)
        pin_p = torch.from_numpy(np.frombuffer(t, np.uint8).copy()).pin_memory(); pin_r = torch.from_numpy(np.frombuffer(rrng, np.uint8).copy()).pin_memory()
        p_sk, _k = capi._in(sk, 64)
        i = calls = 0
        while i < n:
            k = min(n - i, r.choice((1, 1, 1, 2, 3, 7, 33, 64))); calls += 1
            how = r.randrange(3)
            if how == 0:
                got = eng.refund(sk, t[pb * i:pb * (i + k)], rrng[128 * i:128 * (i + k)], capi.RNG_PER_LANE)
            elif how == 1:
                o_rf = torch.zeros(128 * k, dtype=torch.uint8).pin_memory(); o_st = torch.full((k,), 9, dtype=torch.uint8).pin_memory()
                eng._ck(eng.lib.act_refund_batch(eng.ctx, k, capi.MEM_HOST, p_sk, pin_p.data_ptr() + pb * i, pin_r.data_ptr() + 128 * i, capi.RNG_PER_LANE, o_rf.data_ptr(), o_st.data_ptr()))
                got = (o_st.numpy().tobytes(), o_rf.numpy().tobytes())
            else:
                k = 1
                got = eng.refund(sk, t[pb * i:pb * (i + 1)], rrng[128 * i:128 * (i + 1)], capi.RNG_SEQUENTIAL)
                so, ro = octx.refund(sk, t[pb * i:pb * (i + 1)], rrng[128 * i:128 * (i + 1)])
                assert got == (bytes([so]), ro), ("one-proof refund vs oracle", L, i)
            assert got == (st2_b[i:i + k], rf_b[128 * i:128 * (i + k)]), ("tiny refund", L, i, k, how)
            i += k
        assert eng.secret_residue() == 0
        tiny_note = "; %d refunds of 1 - 64 proofs (signature beside the verification; pageable / pinned-in-place / sequential one-proof) == the batch answers" % calls
    hist = {}
    for s_ in st_o: hist[s_] = hist.get(s_, 0) + 1
    print("L=%d: %d proofs, statuses %s: engine == oracle (both transcript modes, refunds sampled)%s" % (L, n, dict(sorted(hist.items())), ("; %d concurrent small calls from %d threads == the batch answers" % (len(cuts), T) if T else "") + tiny_note))
    eng.close()
