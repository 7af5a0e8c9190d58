// abi_conformance.cpp — a C++ caller compiled against include/act_mi355x.h and linked to libact_mi355x.so (no ctypes, no
// Python, no torch in the process): replays a golden lifecycle fixture through the hot-path entry points exactly as the
// crate's `mod mi355x` would call them and exits non-zero on the first byte that differs.  The fixture
// arrives as a flat binary written by tests/test_gpu_abi_conformance.py from tests/golden/sodium_lifecycle_L128.json
// (known answers computed by libsodium + LLVM BLAKE3), so this is at once a prototype check of the header (a drifted
// argument order or width does not compile or does not reproduce the bytes) and a parity test of the C ABI itself.
//
//   g++ -std=c++17 -Iinclude tests/abi_conformance.cpp -o conf -Lanonymous-credit-tokens_amd -lact_mi355x ...
//   ./conf fixture.bin
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "act_mi355x.h"

typedef std::vector<uint8_t> bytes;

static FILE* g_f;
static uint64_t rd_u64() { uint64_t v; if (fread(&v, 8, 1, g_f) != 1) { fprintf(stderr, "short fixture\n"); exit(2); } return v; }
static bytes rd_blob() { uint64_t n = rd_u64(); bytes b(n); if (n && fread(b.data(), 1, n, g_f) != n) { fprintf(stderr, "short fixture\n"); exit(2); } return b; }

static int g_fail = 0;
static void expect(const char* what, const bytes& got, const bytes& want) {
  if (got.size() != want.size() || memcmp(got.data(), want.data(), got.size()) != 0) {
    size_t i = 0; while (i < got.size() && i < want.size() && got[i] == want[i]) i++;
    fprintf(stderr, "MISMATCH %s: first difference at byte %zu of %zu\n", what, i, want.size());
    g_fail++;
  } else {
    printf("ok  %-44s %zu bytes\n", what, got.size());
  }
}
#define CK(call) do { int rc_ = (call); if (rc_ != ACT_OK) { fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, ctx ? act_last_error(ctx) : ""); return 3; } } while (0)

int main(int argc, char** argv) {
  if (argc < 2 || !(g_f = fopen(argv[1], "rb"))) { fprintf(stderr, "usage: %s fixture.bin\n", argv[0]); return 2; }
  const int L = (int)rd_u64();
  const size_t n = (size_t)rd_u64();
  std::vector<std::string> pa;
  for (int i = 0; i < 4; i++) { bytes b = rd_blob(); pa.emplace_back(b.begin(), b.end()); }
  const bytes h = rd_blob(), sk_rng = rd_blob(), sk = rd_blob(), sk_other = rd_blob();
  const bytes pre_rng = rd_blob(), pre = rd_blob(), req_rng = rd_blob(), req = rd_blob(), camt = rd_blob(), issue_rng = rd_blob(), resp = rd_blob();
  const bytes tok = rd_blob(), s = rd_blob(), prove_rng = rd_blob(), proof_made = rd_blob(), prerefund = rd_blob(), proof_in = rd_blob();
  const bytes status = rd_blob(), kprime = rd_blob(), refund_rng = rd_blob(), refund = rd_blob(), tok2 = rd_blob(), status_tok2 = rd_blob(), status_other = rd_blob();
  const bytes seq_refund = rd_blob();       // refunds under ACT_RNG_SEQUENTIAL with refund_rng as the one stream
  fclose(g_f);

  act_ctx* ctx = nullptr;
  bytes out_h(96);
  CK(act_params_new(0, pa[0].c_str(), pa[1].c_str(), pa[2].c_str(), pa[3].c_str(), out_h.data()));
  expect("act_params_new", out_h, h);
  CK(act_ctx_create(h.data(), L, 0, 7, &ctx));                                   // ragged chunks: 7 records per launch
  if (act_spend_proof_bytes(ctx) != 32u * (14 + 4 * (size_t)L) || act_prove_rng_bytes(ctx) != 64u * (4 * (size_t)L + 12)) { fprintf(stderr, "size queries wrong\n"); return 4; }
  const size_t pb = act_spend_proof_bytes(ctx);

  for (int mode = ACT_TRANSCRIPT_HOST; mode <= ACT_TRANSCRIPT_DEVICE; mode++) {
    printf("-- transcript mode %d\n", mode);
    CK(act_ctx_set_transcript_mode(ctx, mode));
    bytes o_sk(64); CK(act_private_key_random(ctx, sk_rng.data(), o_sk.data())); expect("act_private_key_random", o_sk, sk);
    bytes o_pre(n * 64); CK(act_pre_issuance_random_batch(ctx, n, ACT_MEM_HOST, pre_rng.data(), o_pre.data())); expect("act_pre_issuance_random_batch", o_pre, pre);
    bytes o_req(n * 128); CK(act_request_batch(ctx, n, ACT_MEM_HOST, pre.data(), req_rng.data(), o_req.data())); expect("act_request_batch", o_req, req);
    bytes o_resp(n * 160), st(n);
    CK(act_issue_batch(ctx, n, ACT_MEM_HOST, sk.data(), req.data(), camt.data(), issue_rng.data(), ACT_RNG_PER_LANE, o_resp.data(), st.data()));
    expect("act_issue_batch", o_resp, resp); expect("act_issue_batch status", st, bytes(n, 0));
    bytes o_tok(n * 160);
    CK(act_issuance_to_credit_token_batch(ctx, n, ACT_MEM_HOST, pre.data(), sk.data() + 32, req.data(), resp.data(), o_tok.data(), st.data()));
    expect("act_issuance_to_credit_token_batch", o_tok, tok); expect("  status", st, bytes(n, 0));
    bytes o_proof(n * pb), o_prer(n * 96);
    CK(act_prove_spend_batch(ctx, n, ACT_MEM_HOST, tok.data(), s.data(), prove_rng.data(), o_proof.data(), o_prer.data(), st.data()));
    expect("act_prove_spend_batch proofs", o_proof, proof_made); expect("act_prove_spend_batch prerefunds", o_prer, prerefund);
    bytes o_kp(n * 32);
    CK(act_verify_spend_batch(ctx, n, ACT_MEM_HOST, sk.data(), proof_in.data(), st.data(), o_kp.data()));
    expect("act_verify_spend_batch status", st, status); expect("act_verify_spend_batch K'", o_kp, kprime);
    CK(act_verify_spend_batch(ctx, n, ACT_MEM_HOST, sk_other.data(), proof_in.data(), st.data(), nullptr));
    expect("act_verify_spend_batch (other issuer)", st, status_other);
    bytes o_rf(n * 128);
    CK(act_refund_batch(ctx, n, ACT_MEM_HOST, sk.data(), proof_in.data(), refund_rng.data(), ACT_RNG_PER_LANE, o_rf.data(), st.data()));
    expect("act_refund_batch", o_rf, refund); expect("act_refund_batch status", st, status);
    CK(act_refund_batch(ctx, n, ACT_MEM_HOST, sk.data(), proof_in.data(), refund_rng.data(), ACT_RNG_SEQUENTIAL, o_rf.data(), st.data()));
    expect("act_refund_batch (sequential rng)", o_rf, seq_refund);
    bytes o_tok2(n * 160);
    CK(act_refund_to_credit_token_batch(ctx, n, ACT_MEM_HOST, prerefund.data(), proof_in.data(), refund.data(), sk.data() + 32, o_tok2.data(), st.data()));
    expect("act_refund_to_credit_token_batch", o_tok2, tok2); expect("  status", st, status_tok2);
    // the two halves of refund through the split entry points
    CK(act_verify_spend_batch(ctx, n, ACT_MEM_HOST, sk.data(), proof_in.data(), st.data(), o_kp.data()));
    bytes st2(n);
    CK(act_refund_sign_batch(ctx, n, ACT_MEM_HOST, sk.data(), o_kp.data(), st.data(), refund_rng.data(), ACT_RNG_SEQUENTIAL, o_rf.data(), st2.data()));
    expect("act_verify_spend_batch + act_refund_sign_batch", o_rf, seq_refund);
  }

  // the node handle over three contexts on device 0: same bytes, both rng modes
  act_node* node = nullptr;
  const int devs[3] = {0, 0, 0};
  int rc = act_node_create(h.data(), L, devs, 3, 7, &node);
  if (rc) { fprintf(stderr, "act_node_create -> %d (%s)\n", rc, act_node_last_error(node)); return 3; }
  bytes st(n), o_rf(n * 128), o_kp(n * 32);
  rc = act_node_verify_spend_batch(node, n, sk.data(), proof_in.data(), st.data(), o_kp.data());
  if (rc) { fprintf(stderr, "act_node_verify_spend_batch -> %d (%s)\n", rc, act_node_last_error(node)); return 3; }
  expect("act_node_verify_spend_batch status", st, status); expect("act_node_verify_spend_batch K'", o_kp, kprime);
  rc = act_node_refund_batch(node, n, sk.data(), proof_in.data(), refund_rng.data(), ACT_RNG_PER_LANE, o_rf.data(), st.data());
  if (rc) { fprintf(stderr, "act_node_refund_batch -> %d (%s)\n", rc, act_node_last_error(node)); return 3; }
  expect("act_node_refund_batch", o_rf, refund);
  rc = act_node_refund_batch(node, n, sk.data(), proof_in.data(), refund_rng.data(), ACT_RNG_SEQUENTIAL, o_rf.data(), st.data());
  if (rc) { fprintf(stderr, "act_node_refund_batch -> %d (%s)\n", rc, act_node_last_error(node)); return 3; }
  expect("act_node_refund_batch (sequential rng)", o_rf, seq_refund);
  // wire bytes in, wire bytes out: the proofs as SpendProof::to_cbor would send them, the refunds as
  // Refund::to_cbor frames them -- sequential bytes, then the generator itself as a draw callback (what rust/src/mi355x.rs passes)
  {
    const size_t ml = act_cbor_size(ctx, ACT_CBOR_SPEND_PROOF), rl = act_cbor_size(ctx, ACT_CBOR_REFUND);
    if (rl != 141) { fprintf(stderr, "Refund message size %zu\n", rl); return 4; }
    bytes msgs(n * ml);
    CK(act_cbor_encode_batch(ctx, ACT_CBOR_SPEND_PROOF, n, ACT_MEM_HOST, proof_in.data(), msgs.data()));
    bytes want(n * rl, 0);
    for (size_t i = 0; i < n; i++) {
      if (status[i]) continue;
      uint8_t* m = want.data() + i * rl;
      *m++ = 0xa4;
      for (int f = 0; f < 4; f++) { *m++ = (uint8_t)(f + 1); *m++ = 0x58; *m++ = 0x20; memcpy(m, seq_refund.data() + i * 128 + 32 * f, 32); m += 32; }
    }
    bytes o_msgs(n * rl, 7);
    rc = act_node_refund_cbor_batch(node, n, sk.data(), msgs.data(), nullptr, refund_rng.data(), ACT_RNG_SEQUENTIAL, o_msgs.data(), st.data());
    if (rc) { fprintf(stderr, "act_node_refund_cbor_batch -> %d (%s)\n", rc, act_node_last_error(node)); return 3; }
    expect("act_node_refund_cbor_batch (sequential rng)", o_msgs, want); expect("  status", st, status);
    struct Gen { const bytes* stream; size_t pos; int draws; } gen{&refund_rng, 0, 0};
    act_rng_source src{[](void* g_, uint8_t* dst, size_t len) -> int { Gen* g = static_cast<Gen*>(g_); if (g->pos + len > g->stream->size()) return 1; memcpy(dst, g->stream->data() + g->pos, len); g->pos += len; g->draws++; return 0; }, &gen};
    std::fill(o_msgs.begin(), o_msgs.end(), 7);
    rc = act_node_refund_cbor_batch(node, n, sk.data(), msgs.data(), nullptr, reinterpret_cast<const uint8_t*>(&src), ACT_RNG_CALLBACK, o_msgs.data(), st.data());
    if (rc) { fprintf(stderr, "act_node_refund_cbor_batch (callback) -> %d (%s)\n", rc, act_node_last_error(node)); return 3; }
    expect("act_node_refund_cbor_batch (generator callback)", o_msgs, want);
    size_t accepted = 0; for (size_t i = 0; i < n; i++) accepted += status[i] == 0;
    if (gen.draws != 1 || gen.pos != 128 * accepted) { fprintf(stderr, "callback drew %zu bytes in %d calls, want %zu in 1\n", gen.pos, gen.draws, 128 * accepted); g_fail++; }
    else printf("ok  the generator was asked once, for 128 bytes per accepted message (%zu)\n", gen.pos);
    // the redemption step on wire bytes: every accepted nullifier is fresh the first time, spent the second
    act_node_nullifier_set* nset = nullptr;
    const int ndev[2] = {0, 0};
    rc = act_node_nullifier_set_create(ndev, 2, 1024, nullptr, &nset);
    if (rc) { fprintf(stderr, "act_node_nullifier_set_create -> %d\n", rc); return 3; }
    gen = Gen{&refund_rng, 0, 0};
    rc = act_node_redeem_cbor_batch(node, nset, n, sk.data(), msgs.data(), nullptr, reinterpret_cast<const uint8_t*>(&src), ACT_RNG_CALLBACK, o_msgs.data(), st.data());
    if (rc) { fprintf(stderr, "act_node_redeem_cbor_batch -> %d (%s)\n", rc, act_node_last_error(node)); return 3; }
    expect("act_node_redeem_cbor_batch (first submission)", o_msgs, want); expect("  status", st, status);
    if (act_node_nullifier_set_len(nset) != accepted) { fprintf(stderr, "nullifier set holds %zu, want %zu\n", act_node_nullifier_set_len(nset), accepted); g_fail++; }
    gen = Gen{&refund_rng, 0, 0};
    rc = act_node_redeem_cbor_batch(node, nset, n, sk.data(), msgs.data(), nullptr, reinterpret_cast<const uint8_t*>(&src), ACT_RNG_CALLBACK, o_msgs.data(), st.data());
    bytes spent = status; for (size_t i = 0; i < n; i++) if (!spent[i]) spent[i] = ACT_STATUS_DOUBLE_SPEND;
    if (rc) { fprintf(stderr, "act_node_redeem_cbor_batch (second) -> %d\n", rc); return 3; }
    expect("act_node_redeem_cbor_batch (second submission: double spends)", st, spent); expect("  no refunds", o_msgs, bytes(n * rl, 0));
    if (gen.pos != 0) { fprintf(stderr, "a rejected batch drew %zu bytes\n", gen.pos); g_fail++; }
    act_node_nullifier_set_destroy(nset);
    // one message per call with its 128 bytes, a server's own rhythm: the refund is one call (signature beside the verification), then
    // the store decides -- message by message the same bytes as the batch above, and the same store
    rc = act_node_nullifier_set_create(ndev, 2, 1024, nullptr, &nset);
    if (rc) { fprintf(stderr, "act_node_nullifier_set_create -> %d\n", rc); return 3; }
    bytes one_by_one(n * rl, 7), st1(n, 9);
    size_t cur = 0;
    for (size_t i = 0; i < n; i++) {
      rc = act_node_redeem_cbor_batch(node, nset, 1, sk.data(), msgs.data() + i * ml, nullptr, refund_rng.data() + 128 * cur, ACT_RNG_SEQUENTIAL, one_by_one.data() + i * rl, st1.data() + i);
      if (rc) { fprintf(stderr, "act_node_redeem_cbor_batch (message %zu) -> %d (%s)\n", i, rc, act_node_last_error(node)); return 3; }
      cur += st1[i] == 0;
    }
    expect("act_node_redeem_cbor_batch, one message per call", one_by_one, want); expect("  status", st1, status);
    if (act_node_nullifier_set_len(nset) != accepted) { fprintf(stderr, "nullifier set holds %zu, want %zu\n", act_node_nullifier_set_len(nset), accepted); g_fail++; }
    act_node_nullifier_set_destroy(nset);
    // refund + nullifiers in one call (a caller with a store of its own): the nullifier is the first field of the record
    bytes nul(n * 32, 7), keyed(n * rl, 7);
    rc = act_refund_cbor_keys_batch(ctx, n, ACT_MEM_HOST, sk.data(), msgs.data(), nullptr, refund_rng.data(), ACT_RNG_SEQUENTIAL, keyed.data(), st.data(), nul.data());
    if (rc) { fprintf(stderr, "act_refund_cbor_keys_batch -> %d (%s)\n", rc, act_last_error(ctx)); return 3; }
    expect("act_refund_cbor_keys_batch", keyed, want); expect("  status", st, status);
    bytes first_fields(n * 32);
    for (size_t i = 0; i < n; i++) memcpy(first_fields.data() + 32 * i, proof_in.data() + i * (proof_in.size() / n), 32);
    expect("  nullifiers", nul, first_fields);
  }
  act_node_destroy(node);

  // failure path from C: a context is returned for its error text and must be destroyed by the caller
  bytes bad_h = h; bad_h[0] |= 1;
  act_ctx* bad = nullptr;
  rc = act_ctx_create(bad_h.data(), L, 0, 4, &bad);
  if (rc != ACT_ERR_PARAMS || !bad || !*act_last_error(bad)) { fprintf(stderr, "bad params: rc %d\n", rc); g_fail++; } else printf("ok  act_ctx_create rejects a non-canonical h1: \"%s\"\n", act_last_error(bad));
  act_ctx_destroy(bad);
  act_ctx_destroy(ctx);
  if (g_fail) { fprintf(stderr, "%d mismatches\n", g_fail); return 1; }
  printf("conformance: all entry points reproduce the fixture\n");
  return 0;
}
