"""BASELINE.json's configurations at their full single-GPU sizes, checked for correctness (not only timed):
  config 2  2^16 spend-proof verifies at L = 64, HBM-resident, 1/1024 lanes tampered: statuses exact, the first 256 lanes
            and every tampered lane re-verified by the oracle
  config 3  2^20 prove_spend at L = 128 in 2^16-lane chunks with device-resident rng: every proof fed back through the
            verifier (all accepted), 512 sampled proofs byte-compared with the oracle's proof for the same token / rng
  config 4  per-GPU share of "2^22 issue + refund over 8 GPUs": ONE 2^19-lane act_node_issue_batch and ONE 2^19-lane
            act_node_refund_batch through a node handle over two contexts, ACT_RNG_SEQUENTIAL (one rng stream, drawn from only by
            accepted lanes), ~1/512 lanes rejected on every shard; statuses exact, rejected records zero, and the oracle fed
            the slice a sequential loop would have handed each lane on the lanes behind shard boundaries / rejections + random lanes
  config 5  2^20 full lifecycles (request -> issue -> token -> prove_spend -> refund -> token) streamed through PINNED
            HOST memory in 2^16-lane chunks over two contexts working concurrently (the node handle: one context moves
            and hashes while the other computes); every final balance checked
Rates are written to gpurun_out/ when ACT_WRITE_RATES is set (tools/profile_round.sh copies them into profiles/)."""
import json
import os
import time

import pytest

from conftest import ELL, ROOT, shake, scb

pytestmark = pytest.mark.gpu


def note_rate(key, value):
    if not os.environ.get("ACT_WRITE_RATES"):
        return
    path = os.path.join(ROOT, "gpurun_out", os.environ["ACT_WRITE_RATES"])
    os.makedirs(os.path.dirname(path), exist_ok=True)
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[key] = value
    json.dump(d, open(path, "w"), indent=1)


def make_tokens(eng, sk, D, L, tag):
    """D tokens with credits uniform in [0, 2^L) (config 2 c in [0, 2^64), config 3 c in [0, 2^128))."""
    import random
    r = random.Random(tag)
    pre = eng.pre_issuance_random(shake(tag + "-pre", 128 * D))
    req = eng.request(pre, shake(tag + "-rq", 128 * D))
    cs = [r.getrandbits(L) for _ in range(D)]
    cs[0], cs[1] = 2 ** L - 1, 0                            # the extremes of the range
    st, resp = eng.issue(sk, req, b"".join(scb(c) for c in cs), shake(tag + "-ir", 128 * D))
    assert st == bytes(D)
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    assert st == bytes(D)
    return tok, cs


def test_config2_2_16_verifies_at_L64(engine_factory, oracle, bench_params):
    import numpy as np
    import torch
    from act_amd import capi
    L, D, n = 64, 1024, 1 << 16
    eng = engine_factory(bench_params, L, max_batch=0, transcript=capi.TRANSCRIPT_DEVICE)
    sk = eng.private_key_random(shake("c2-sk", 64))
    tok, cs = make_tokens(eng, sk, D, L, "c2")
    import random
    r = random.Random("c2-s")
    ss = [r.randrange(c + 1) for c in cs]                     # s uniform in [0, c]
    st, proofs, _ = eng.prove_spend(tok, b"".join(scb(s) for s in ss), shake("c2-pr", eng.prove_rng_bytes * D))
    assert st == bytes(D)
    pb = eng.proof_bytes
    assert pb == 8640
    dev = torch.from_numpy(np.frombuffer(proofs, np.uint8).reshape(D, pb).copy()).cuda().repeat(n // D, 1).contiguous()
    idx = torch.arange(513, n, 1024, device="cuda")
    dev[idx[0::2], 32] ^= 1                   # charge s
    dev[idx[1::2], 64:96] = 0                 # A' = identity
    status = torch.full((n,), 99, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t = time.perf_counter(); eng.verify_spend_dev(sk, n, dev.data_ptr(), status.data_ptr()); torch.cuda.synchronize(); dt = time.perf_counter() - t
    exp = torch.zeros(n, dtype=torch.uint8, device="cuda"); exp[idx[0::2]] = 7; exp[idx[1::2]] = 6
    assert torch.equal(status, exp)
    # the oracle on the first 256 lanes and on every tampered lane
    lanes = list(range(256)) + idx.cpu().tolist()
    host = dev[torch.tensor(lanes, device="cuda")].cpu().numpy().tobytes()
    octx = oracle.ctx(bench_params, L)
    st_o = octx.verify_spend_batch(sk, host, 8)
    assert list(st_o) == status[torch.tensor(lanes, device="cuda")].cpu().tolist()
    note_rate("config2_verify_L64_2^16", {"verifies_per_s": n / dt, "ms": 1e3 * dt, "note": "first (cold) call"})


def test_config3_2_20_prove_spend_at_L128(engine_factory, oracle, bench_params):
    import numpy as np
    import torch
    from act_amd import capi
    L, chunk, nchunks, sample = 128, 1 << 16, 16, 32
    eng = engine_factory(bench_params, L, max_batch=0, transcript=capi.TRANSCRIPT_DEVICE)
    sk = eng.private_key_random(shake("c3-sk", 64))
    D = 4096
    tok, cs = make_tokens(eng, sk, D, L, "c3")
    import random
    r = random.Random("c3-s")
    ss = [r.randrange(c + 1) for c in cs]                     # s uniform in [0, c]
    ss[5] = cs[5]; ss[6] = 0; ss[0] = cs[0] // 2             # spend everything / nothing / half of 2^128 - 1
    s_b = b"".join(scb(s) for s in ss)
    pb, rb = eng.proof_bytes, eng.prove_rng_bytes
    d_tok = torch.from_numpy(np.frombuffer(tok, np.uint8).reshape(D, 160).copy()).cuda().repeat(chunk // D, 1).contiguous()
    d_s = torch.from_numpy(np.frombuffer(s_b, np.uint8).reshape(D, 32).copy()).cuda().repeat(chunk // D, 1).contiguous()
    d_proof = torch.empty((chunk, pb), dtype=torch.uint8, device="cuda"); d_prer = torch.empty((chunk, 96), dtype=torch.uint8, device="cuda")
    d_st = torch.empty(chunk, dtype=torch.uint8, device="cuda"); d_vst = torch.empty(chunk, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(2024)
    octx = oracle.ctx(bench_params, L)
    t_prove = 0.0
    picked = []
    for c in range(nchunks):
        d_rng = torch.randint(0, 256, (chunk, rb), dtype=torch.uint8, device="cuda", generator=g)
        torch.cuda.synchronize()
        t = time.perf_counter()
        eng.prove_spend_dev(chunk, d_tok.data_ptr(), d_s.data_ptr(), d_rng.data_ptr(), d_proof.data_ptr(), d_prer.data_ptr(), d_st.data_ptr())
        torch.cuda.synchronize(); t_prove += time.perf_counter() - t
        assert int(d_st.sum()) == 0
        eng.verify_spend_dev(sk, chunk, d_proof.data_ptr(), d_vst.data_ptr()); torch.cuda.synchronize()
        assert int(d_vst.sum()) == 0, "every GPU-made proof must verify (chunk %d)" % c
        lanes = torch.tensor([(c * 977 + k * 2039) % chunk for k in range(sample)], device="cuda")
        picked.append((lanes.cpu().tolist(), d_rng[lanes].cpu().numpy().tobytes(), d_proof[lanes].cpu().numpy().tobytes(), d_prer[lanes].cpu().numpy().tobytes()))
        del d_rng
    # 512 sampled proofs against the oracle's proof for the same token, charge and rng bytes
    for lanes, rng, got_p, got_r in picked:
        toks = b"".join(tok[160 * (i % D):160 * (i % D) + 160] for i in lanes)
        sb = b"".join(s_b[32 * (i % D):32 * (i % D) + 32] for i in lanes)
        want_p, want_r = octx.prove_spend_batch(toks, sb, rng, 8)
        assert got_p == want_p and got_r == want_r
    note_rate("config3_prove_spend_L128_2^20", {"proofs_per_s": nchunks * chunk / t_prove, "ms": 1e3 * t_prove, "chunks": nchunks})


@pytest.mark.parametrize("rng_source,nchunks,streams", [("seeded", 32, 1), ("seeded", 32, 2), ("bytes", 8, 1)])
def test_config5_lifecycles_streamed_from_pinned_host_memory(bench_params, oracle, rng_source, nchunks, streams):
    """BASELINE configs[4] at ONE GPU's share: 2^24 lifecycles over 8 GPUs = 2^21 per GPU, streamed in 2^16-lane calls through pinned
    host memory (request -> issue -> to_credit_token -> prove_spend -> refund -> to_credit_token), every call a node-handle call.
    "seeded": the prover's generators are seeded (act_node_prove_spend_seeded_batch: BLAKE3-XOF(seed | lane) expanded in HBM), so
    the 33 536 rng bytes per proof never cross PCIe; "bytes" (2^19 lifecycles): every rng byte comes from host memory, the
    round-3 form.  streams = 2: the configuration's "double-buffered" -- two host threads, a node handle and a set of buffers each,
    every other 2^16-lane chunk each: one stream's copies cross PCIe while the other's kernels run.  Balances of every lane; four
    lanes of each stream's last call byte for byte against the oracle's whole lifecycle."""
    import threading
    import numpy as np
    import torch
    import ctypes as C
    from act_amd import capi
    L, chunk = 128, 1 << 16
    eng0 = capi.Engine(bench_params, 8, max_batch=4)
    sk = eng0.private_key_random(shake("c5-sk", 64)); eng0.close()
    pin = lambda *shape: torch.empty(shape, dtype=torch.uint8, pin_memory=True)
    seed = shake("c5-seed", 32)
    amounts = np.array([(i * 40503 + 11) % 100000 + 1 for i in range(chunk)], dtype=np.uint64)
    charges = amounts // np.uint64(3)
    le32 = lambda v: np.concatenate([v.astype("<u8").view(np.uint8).reshape(-1, 8), np.zeros((len(v), 24), np.uint8)], axis=1)
    c_b, s_b, m_b = le32(amounts).tobytes(), le32(charges).tobytes(), le32(amounts - charges)
    cb = np.frombuffer(c_b, np.uint8); sb = np.frombuffer(s_b, np.uint8)
    skb = (C.c_uint8 * 64).from_buffer_copy(sk); wb = (C.c_uint8 * 32).from_buffer_copy(sk[32:])
    seedb = (C.c_uint8 * 32).from_buffer_copy(seed)
    ptr = lambda t: t.data_ptr()
    spent = {}          # where a streamed lifecycle's time goes: seconds per call kind, summed over the chunks (and streams)
    lock = threading.Lock()
    errors = []

    class Stream:
        def __init__(self, tid):
            self.tid = tid
            # one stream: the node handle cuts every call over two contexts; two streams: a context each
            self.node = capi.Node(bench_params, L, devices=(0, 0) if streams == 1 else (0,), max_batch=1 << 14, transcript=capi.TRANSCRIPT_DEVICE)
            self.pb, self.rb = self.node.proof_bytes, self.node.prove_rng_bytes
            self.g = torch.Generator(device="cuda"); self.g.manual_seed(5 + tid)
            self.bufs = {k: pin(chunk, v) for k, v in dict(pre=64, req=128, resp=160, tok=160, proof=self.pb, prer=96, rf=128, tok2=160).items()}
            self.r128 = {k: pin(chunk, 128) for k in ("pre", "rq", "ir", "rr")}
            self.st = pin(chunk)
            self.last = None
            if rng_source == "bytes":
                # the 2.2 GB of prover bytes are drawn once and reused by every chunk (tokens differ per chunk)
                self.r_pr = pin(chunk, self.rb); self.r_pr.copy_(self.rnd(chunk, self.rb))

        def rnd(self, *shape):
            return torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=self.g)

        def timed(self, key, fn):
            t = time.perf_counter(); r = fn(); d = time.perf_counter() - t
            with lock:
                spent[key] = spent.get(key, 0.0) + d
            return r

        def run(self, chunks):
            try:
                self._run(chunks)
            except BaseException as e:          # a failed assertion in a worker thread must fail the test
                errors.append(e)

        def _run(self, chunks):
            node, bufs, r128, st, timed = self.node, self.bufs, self.r128, self.st, self.timed
            lib, nd, ck = node.lib, node.nd, node._ck
            # PreIssuance::random has no node-level twin (it is two scalar reductions): context 0 of the node does it
            ctx0 = lib.act_node_ctx(nd, 0)
            for c in chunks:
                def draw():
                    for k in r128:
                        r128[k].copy_(self.rnd(chunk, 128))
                    torch.cuda.current_stream().synchronize()
                timed("draw 4 x 128 rng bytes per lane (torch, D2H)", draw)
                assert timed("pre_issuance_random", lambda: lib.act_pre_issuance_random_batch(ctx0, chunk, capi.MEM_HOST, ptr(r128["pre"]), ptr(bufs["pre"]))) == 0
                ck(timed("request", lambda: lib.act_node_request_batch(nd, chunk, ptr(bufs["pre"]), ptr(r128["rq"]), ptr(bufs["req"]))))
                ck(timed("issue", lambda: lib.act_node_issue_batch(nd, chunk, skb, ptr(bufs["req"]), cb.ctypes.data, ptr(r128["ir"]), capi.RNG_PER_LANE, ptr(bufs["resp"]), ptr(st))))
                assert int(st.sum()) == 0
                ck(timed("issuance_to_credit_token", lambda: lib.act_node_issuance_to_credit_token_batch(nd, chunk, ptr(bufs["pre"]), wb, ptr(bufs["req"]), ptr(bufs["resp"]), ptr(bufs["tok"]), ptr(st))))
                assert int(st.sum()) == 0
                if rng_source == "seeded":
                    ck(timed("prove_spend", lambda: lib.act_node_prove_spend_seeded_batch(nd, chunk, ptr(bufs["tok"]), sb.ctypes.data, seedb, C.c_uint64(c * chunk), ptr(bufs["proof"]), ptr(bufs["prer"]), ptr(st))))
                else:
                    ck(timed("prove_spend", lambda: lib.act_node_prove_spend_batch(nd, chunk, ptr(bufs["tok"]), sb.ctypes.data, ptr(self.r_pr), ptr(bufs["proof"]), ptr(bufs["prer"]), ptr(st))))
                assert int(st.sum()) == 0
                cut_prove = (node.device_stats(), node.balance_state())
                ck(timed("refund", lambda: lib.act_node_refund_batch(nd, chunk, skb, ptr(bufs["proof"]), ptr(r128["rr"]), capi.RNG_PER_LANE, ptr(bufs["rf"]), ptr(st))))
                if int(st.sum()) != 0:      # say where: which lanes, and how the dispatcher had cut the call
                    bad = np.nonzero(st.numpy())[0]
                    raise AssertionError("every honest spend must be refunded (chunk %d): %d lanes rejected, first %d last %d, statuses %s; cut of the refund: %s %s; of prove_spend: %s"
                                         % (c, len(bad), bad[0], bad[-1], sorted(set(st.numpy()[bad].tolist())), node.device_stats(), node.balance_state(), cut_prove))
                ck(timed("refund_to_credit_token", lambda: lib.act_node_refund_to_credit_token_batch(nd, chunk, ptr(bufs["prer"]), ptr(bufs["proof"]), ptr(bufs["rf"]), wb, ptr(bufs["tok2"]), ptr(st))))
                assert int(st.sum()) == 0
                # final balances: the new token carries c - s; its nullifier k is the fresh k* of the spend, not the old one
                t2 = bufs["tok2"].numpy()
                assert np.array_equal(t2[:, 128:160], m_b), "balance c - s wrong in chunk %d" % c
                assert np.array_equal(t2[:, 64:96], bufs["prer"].numpy()[:, 32:64])
                assert not np.array_equal(t2[:, 64:96], bufs["tok"].numpy()[:, 64:96])
                self.last = c

    ss = [Stream(t) for t in range(streams)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if streams == 1:
        ss[0].run(range(nchunks))
    else:
        th = [threading.Thread(target=ss[t].run, args=(range(t, nchunks, streams),)) for t in range(streams)]
        for x in th:
            x.start()
        for x in th:
            x.join()
    dt = time.perf_counter() - t0
    if errors:
        raise errors[0]
    # four lanes of every stream's last call, every record of their lifecycle, against the oracle fed the same inputs
    octx = oracle.ctx(bench_params, L)
    for S in ss:
        last = S.last * chunk
        for i in (0, 1, chunk // 2 + 3, chunk - 1):
            row = lambda t, w: t.numpy()[i].tobytes()[:w]
            bufs, r128, pb, rb = S.bufs, S.r128, S.pb, S.rb
            pre = octx.pre_issuance_random(row(r128["pre"], 128)); assert pre == row(bufs["pre"], 64)
            req = octx.request(pre, row(r128["rq"], 128)); assert req == row(bufs["req"], 128)
            so, resp = octx.issue(sk, req, c_b[32 * i:32 * i + 32], row(r128["ir"], 128)); assert so == 0 and resp == row(bufs["resp"], 160)
            so, tok = octx.issuance_to_credit_token(pre, sk[32:], req, resp); assert so == 0 and tok == row(bufs["tok"], 160)
            prng = oracle.blake3(seed + (last + i).to_bytes(8, "little"), rb) if rng_source == "seeded" else row(S.r_pr, rb)
            so, proof, prer = octx.prove_spend(tok, s_b[32 * i:32 * i + 32], prng); assert so == 0 and proof == row(bufs["proof"], pb) and prer == row(bufs["prer"], 96)
            so, rf = octx.refund(sk, proof, row(r128["rr"], 128)); assert so == 0 and rf == row(bufs["rf"], 128)
            so, tok2 = octx.refund_to_credit_token(prer, proof, rf, sk[32:]); assert so == 0 and tok2 == row(bufs["tok2"], 160)
        S.node.close()
    note_rate("config5_lifecycles_L128_2^%d_streamed_pinned_host_%s_rng%s" % ((nchunks * chunk).bit_length() - 1, rng_source, "" if streams == 1 else "_%d_streams" % streams),
              {"lifecycles_per_s": nchunks * chunk / dt, "ms": 1e3 * dt, "lifecycles": nchunks * chunk, "host_streams": streams,
               "ms_per_2^16_lanes_by_call": {k: round(1e3 * v / nchunks, 2) for k, v in spent.items()},
               "ms_per_2^16_lanes_python_checks": round(1e3 * (streams * dt - sum(spent.values())) / nchunks, 2),
               "note": ("2^16-lane calls through pinned host memory, two contexts on one GPU (node handle), device transcripts; " if streams == 1 else
                        "double-buffered: two host threads with a one-context node handle and buffers each, 2^16-lane calls through pinned host memory, device transcripts "
                        "(per-call times are each thread's own wall time: they overlap); ")
                       + "includes drawing the 128-byte rng slices; "
                       + ("prover generators seeded (BLAKE3-XOF expanded in HBM)" if rng_source == "seeded" else "prover rng bytes (33 536 per proof) from host memory")})


def _accepted_index(status):
    """index of every lane among the accepted lanes in front of it = its 128-byte slice of a sequential rng stream
    (/root/reference/src/lib.rs:638-643, 842-846: the generator is touched only after the proof verified)"""
    import numpy as np
    ok = (np.frombuffer(status, np.uint8) == 0)
    return np.cumsum(ok) - ok


def _sample_lanes(n, parts, status, extra_random, seed):
    """lanes whose rng slice depends on what happened in EARLIER shards (the first lanes of every later shard, the lanes behind
    the first rejections of every shard) + random lanes"""
    import random
    import numpy as np
    st = np.frombuffer(status, np.uint8)
    lanes = set()
    for k in range(parts):
        a, b = n * k // parts, n * (k + 1) // parts
        lanes.update(range(a, min(b, a + 8)))
        rej = np.nonzero(st[a:b])[0][:8] + a
        for r in rej:
            lanes.update(x for x in (r - 1, r, r + 1, r + 2) if 0 <= x < n)
        lanes.update(range(max(a, b - 4), b))
    r = random.Random(seed)
    lanes.update(r.randrange(n) for _ in range(extra_random))
    return sorted(lanes)


def test_config4_2_19_issue_and_refund_per_gpu_sequential_rng(oracle, bench_params):
    import numpy as np
    import torch
    from act_amd import capi
    import bench
    L, n, parts = 128, 1 << 19, 2
    PB = 32 * (14 + 4 * L)
    eng = capi.Engine(bench_params, L, max_batch=1 << 14, transcript=capi.TRANSCRIPT_DEVICE)
    sk = eng.private_key_random(shake("c4-sk", 64))
    octx = oracle.ctx(bench_params, L)
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    rnd = lambda *shape: torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=g)
    # ---- inputs: 2^19 valid requests with c uniform in [1, 2^32), 2^19 distinct valid proofs -------------------------
    pre = torch.empty((n, 64), dtype=torch.uint8, device="cuda"); req = torch.empty((n, 128), dtype=torch.uint8, device="cuda")
    r0, r1 = rnd(n, 128), rnd(n, 128)
    torch.cuda.synchronize()
    eng.pre_issuance_random_dev(n, r0.data_ptr(), pre.data_ptr())
    eng.request_dev(n, pre.data_ptr(), r1.data_ptr(), req.data_ptr())
    del r0, r1
    camt = np.zeros((n, 32), np.uint8)
    camt[:, :4] = np.random.default_rng(4).integers(1, 2 ** 32, n, dtype=np.uint64).astype("<u4").view(np.uint8).reshape(n, 4)
    proofs_dev, _ = bench.make_distinct_proofs_on_device(eng, capi, torch, np, sk, n, L, 4, 1 << 14)
    eng.close()
    # ---- ~1/512 rejected lanes, all kinds, on every shard -----------------------------------------------------------------
    bad = np.arange(257, n, 512)
    reqs = req.cpu().numpy(); del req, pre
    exp_issue = np.zeros(n, np.uint8)
    for t, i in enumerate(bad):
        kind = t % 3
        if kind == 0: reqs[i, 32] ^= 1; exp_issue[i] = 1                      # gamma  -> InvalidIssuanceRequestProof
        elif kind == 1: reqs[i, 0:32] = 0xff; exp_issue[i] = 255              # K undecodable
        else: reqs[i, 64 + (t % 32)] ^= 0x10; exp_issue[i] = 1                # k_bar
    proofs = proofs_dev.cpu().numpy(); del proofs_dev
    torch.cuda.empty_cache()
    exp_refund = np.zeros(n, np.uint8)
    for t, i in enumerate(bad):
        kind = t % 4
        if kind == 0: proofs[i, 32] ^= 1; exp_refund[i] = 7                   # charge s -> InvalidClientSpendProof
        elif kind == 1: proofs[i, 64:96] = 0; exp_refund[i] = 6               # A' = identity -> IdentityPointError
        elif kind == 2: proofs[i, 32 * (4 + L)] ^= 2; exp_refund[i] = 7       # gamma
        else: proofs[i, 32 * (4 + 5):32 * (4 + 6)] = 0xff; exp_refund[i] = 255   # Com_5 undecodable
    n_acc_i, n_acc_r = int((exp_issue == 0).sum()), int((exp_refund == 0).sum())
    rng_i = np.frombuffer(shake("c4-issue-rng", 128 * 1024), np.uint8)
    rng_i = np.resize(rng_i, (n_acc_i, 128)).copy(); rng_i[:, :4] = np.arange(n_acc_i, dtype="<u4").view(np.uint8).reshape(-1, 4)   # every slice distinct
    rng_r = np.resize(np.frombuffer(shake("c4-refund-rng", 128 * 1024), np.uint8), (n_acc_r, 128)).copy()
    rng_r[:, :4] = np.arange(n_acc_r, dtype="<u4").view(np.uint8).reshape(-1, 4)

    node = capi.Node(bench_params, L, devices=(0,) * parts, max_batch=1 << 14, transcript=capi.TRANSCRIPT_DEVICE)
    try:
        lib, nd = node.lib, node.nd
        import ctypes as C
        skb = (C.c_uint8 * 64).from_buffer_copy(sk)
        # ---- issue: one 2^19-lane call ----------------------------------------------------------------------------------
        resp = np.zeros((n, 160), np.uint8); st = np.full(n, 99, np.uint8)
        t = time.perf_counter()
        node._ck(lib.act_node_issue_batch(nd, n, skb, reqs.ctypes.data, camt.ctypes.data, rng_i.ctypes.data, capi.RNG_SEQUENTIAL, resp.ctypes.data, st.ctypes.data))
        dt_issue = time.perf_counter() - t
        assert np.array_equal(st, exp_issue)
        assert not resp[exp_issue != 0].any() and resp[exp_issue == 0].any(axis=1).all()
        slot = _accepted_index(st.tobytes())
        lanes = _sample_lanes(n, parts, st.tobytes(), 8192, 41)
        n_issue_checked = len(lanes)
        acc = [i for i in lanes if exp_issue[i] == 0]
        st_o, resp_o = octx.issue_batch(sk, reqs[acc].tobytes(), camt[acc].tobytes(), rng_i[slot[acc]].tobytes(), 8)
        assert st_o == bytes(len(acc)) and resp_o == resp[acc].tobytes(), "issue: node != oracle on the sequential slices"
        rej = [i for i in lanes if exp_issue[i] != 0]
        st_o, _ = octx.issue_batch(sk, reqs[rej].tobytes(), camt[rej].tobytes(), bytes(128 * len(rej)), 8)
        assert st_o == exp_issue[rej].tobytes()
        # ---- refund: one 2^19-lane call ---------------------------------------------------------------------------------
        rf = np.zeros((n, 128), np.uint8); st = np.full(n, 99, np.uint8)
        t = time.perf_counter()
        node._ck(lib.act_node_refund_batch(nd, n, skb, proofs.ctypes.data, rng_r.ctypes.data, capi.RNG_SEQUENTIAL, rf.ctypes.data, st.ctypes.data))
        dt_refund = time.perf_counter() - t
        assert np.array_equal(st, exp_refund)
        assert not rf[exp_refund != 0].any() and rf[exp_refund == 0].any(axis=1).all()
        slot = _accepted_index(st.tobytes())
        lanes = _sample_lanes(n, parts, st.tobytes(), 512, 42)
        acc = [i for i in lanes if exp_refund[i] == 0]
        st_o, rf_o = octx.refund_batch(sk, proofs[acc].tobytes(), rng_r[slot[acc]].tobytes(), 8)
        assert st_o == bytes(len(acc)) and rf_o == rf[acc].tobytes(), "refund: node != oracle on the sequential slices"
        rej = [i for i in lanes if exp_refund[i] != 0]
        st_o, _ = octx.refund_batch(sk, proofs[rej].tobytes(), bytes(128 * len(rej)), 8)
        assert st_o == exp_refund[rej].tobytes()
    finally:
        node.close()
    note_rate("config4_per_gpu_2^19_issue_refund_sequential_rng", {
        "issues_per_s": n / dt_issue, "refunds_per_s": n / dt_refund, "ms_issue": 1e3 * dt_issue, "ms_refund": 1e3 * dt_refund,
        "rejected_lanes": int(len(bad)), "oracle_checked_issue_lanes": n_issue_checked, "oracle_checked_refund_lanes": len(lanes), "note":
        "one act_node_issue_batch and one act_node_refund_batch of 2^19 lanes from pageable host memory, node = two contexts on one GPU, "
        "ACT_RNG_SEQUENTIAL (two-phase: check on all shards, host prefix count, sign), device transcripts"})
