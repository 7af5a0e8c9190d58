"""ctypes binding of libact_mi355x.so (C ABI: include/act_mi355x.h).  Host side only moves bytes."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACT_LIB_PATH") or os.path.join(_HERE, "libact_mi355x.so")   # override: tuning A/B runs only

MEM_HOST, MEM_DEVICE = 0, 1
RNG_PER_LANE, RNG_SEQUENTIAL, RNG_CALLBACK = 0, 1, 2
TRANSCRIPT_HOST, TRANSCRIPT_DEVICE = 0, 1
_ERRS = {1: "ACT_ERR_ARG", 2: "ACT_ERR_HIP", 3: "ACT_ERR_PARAMS", 4: "ACT_ERR_NO_DEVICE", 5: "ACT_ERR_RNG"}

EXPORTS = [
    "act_params_new", "act_params_random", "act_ctx_create", "act_ctx_destroy", "act_ctx_set_transcript_mode",
    "act_ctx_set_host_threads", "act_host_usable_cpus", "act_host_hash_many", "act_host_parallel_for", "act_host_pool_stats", "act_ctx_streams_overlap", "act_ctx_set_pipeline_depth", "act_ctx_set_small_batch_max", "act_ctx_set_fixed_base_bits", "act_node_set_fixed_base_bits", "act_tuning_set", "act_build_has_ct_secret_tables", "act_ctx_fixed_base_bits", "act_last_error", "act_spend_proof_bytes", "act_prove_rng_bytes",
    "act_spend_transcript_bytes", "act_private_key_random", "act_pre_issuance_random_batch", "act_request_batch",
    "act_issue_batch", "act_issuance_to_credit_token_batch", "act_prove_spend_batch", "act_prove_spend_seeded_batch", "act_node_prove_spend_seeded_batch", "act_verify_spend_batch",
    "act_refund_batch", "act_refund_to_credit_token_batch", "act_debug_last_spend_transcripts", "act_debug_scalarmult_batch", "act_debug_secret_residue", "act_prof_enable",
    "act_prof_reset", "act_prof_kernel_count", "act_prof_kernel_name", "act_prof_get", "act_prof_get_busy", "act_ubench_mad_u64_u32", "act_ubench_random_read", "act_ubench_table_read",
    "act_cbor_size", "act_cbor_record_bytes", "act_cbor_encode_batch", "act_cbor_decode_batch", "act_verify_spend_cbor_batch",
    "act_node_verify_spend_cbor_batch", "act_redeem_batch", "act_node_redeem_batch",
    "act_nullifier_set_create", "act_nullifier_set_destroy", "act_nullifier_set_len", "act_nullifier_set_last_error",
    "act_nullifier_check_and_insert_batch",
    "act_issue_check_batch", "act_issue_sign_batch", "act_refund_sign_batch",
    "act_node_create", "act_node_destroy", "act_node_device_count", "act_node_ctx", "act_node_last_error", "act_node_set_transcript_mode",
    "act_node_set_host_threads", "act_node_request_batch", "act_node_issue_batch", "act_node_issuance_to_credit_token_batch",
    "act_node_prove_spend_batch", "act_node_verify_spend_batch", "act_node_refund_batch", "act_node_refund_to_credit_token_batch",
    "act_node_issue_check_batch", "act_node_issue_sign_batch", "act_node_refund_sign_batch",
    "act_node_nullifier_set_create", "act_node_nullifier_set_destroy", "act_node_nullifier_set_len", "act_node_nullifier_set_last_error",
    "act_node_nullifier_check_and_insert_batch",
    "act_verify_spend_cbor_keys_batch", "act_node_verify_spend_cbor_keys_batch", "act_refund_sign_cbor_batch", "act_refund_cbor_batch", "act_refund_cbor_keys_batch",
    "act_node_refund_sign_cbor_batch", "act_node_refund_cbor_batch", "act_redeem_cbor_batch", "act_node_redeem_cbor_batch",
    "act_ctx_host_hash_stats", "act_ctx_set_tiny_calls", "act_node_set_balance", "act_node_device_stats", "act_node_balance_state", "act_debug_set_slowdown", "act_debug_fail_next_signs",
]
CBOR_TYPES = {"IssuanceRequest": 1, "IssuanceResponse": 2, "SpendProof": 3, "Refund": 4, "PrivateKey": 5, "PublicKey": 6,
              "PreIssuance": 7, "CreditToken": 8, "PreRefund": 9}


class ActError(RuntimeError):
    pass


def build(jobs: int = 8) -> str:
    """Compile every HIP source for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    # `all` = libact_mi355x.so, address-free for every secret (the reference's constant-time posture); `fast` =
    # libact_mi355x_fast.so, the same library with scalar-addressed tables for the CLIENT's secrets (prover, request)
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), f"-j{jobs}", "-s", "all", "fast"], check=True)
    return LIB_PATH


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ActError(f"{LIB_PATH} is missing: build it with act_amd.build() / make -C anonymous-credit-tokens_amd/csrc "
                       "(the HIP engine is the only implementation; there is no CPU fallback)")
    # PyTorch ships its own libamdhip64; if this library initialises HIP first (binding /opt/rocm's copy) a later
    # `import torch` in the same process reports "No HIP GPUs are available".  Loading torch's runtime first makes
    # both use one copy, whichever order the caller imports things in.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, sz, i32, u8p = C.c_void_p, C.c_size_t, C.c_int, C.c_void_p
    lib.act_params_new.argtypes = [i32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, u8p]
    lib.act_params_random.argtypes = [i32, u8p, u8p]
    lib.act_ctx_create.argtypes = [u8p, i32, i32, sz, C.POINTER(vp)]
    lib.act_ctx_destroy.argtypes = [vp]
    lib.act_ctx_destroy.restype = None
    lib.act_ctx_set_transcript_mode.argtypes = [vp, i32]
    lib.act_ctx_set_host_threads.argtypes = [vp, i32]
    lib.act_ctx_set_pipeline_depth.argtypes = [vp, i32]
    lib.act_ctx_set_small_batch_max.argtypes = [vp, sz]
    lib.act_ctx_set_fixed_base_bits.argtypes = [vp, i32, i32]
    lib.act_node_set_fixed_base_bits.argtypes = [vp, i32, i32]
    lib.act_tuning_set.argtypes = [C.c_char_p, C.c_int64]
    lib.act_host_usable_cpus.argtypes = []
    lib.act_host_hash_many.argtypes = [u8p, sz, C.c_uint32, sz, i32, u8p]
    lib.act_host_hash_many.restype = None
    lib.act_host_parallel_for.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t), C.c_void_p]
    lib.act_host_parallel_for.restype = None
    lib.act_host_pool_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    lib.act_host_pool_stats.restype = None
    lib.act_ctx_streams_overlap.argtypes = [vp]
    lib.act_build_has_ct_secret_tables.argtypes = []
    lib.act_ctx_fixed_base_bits.argtypes = [vp, i32]
    lib.act_last_error.argtypes = [vp]
    lib.act_last_error.restype = C.c_char_p
    for f in ("act_spend_proof_bytes", "act_prove_rng_bytes", "act_spend_transcript_bytes"):
        getattr(lib, f).argtypes = [vp]
        getattr(lib, f).restype = sz
    lib.act_private_key_random.argtypes = [vp, u8p, u8p]
    lib.act_pre_issuance_random_batch.argtypes = [vp, sz, i32, u8p, u8p]
    lib.act_request_batch.argtypes = [vp, sz, i32, u8p, u8p, u8p]
    lib.act_issue_batch.argtypes = [vp, sz, i32, u8p, u8p, u8p, u8p, i32, u8p, u8p]
    lib.act_issuance_to_credit_token_batch.argtypes = [vp, sz, i32, u8p, u8p, u8p, u8p, u8p, u8p]
    lib.act_prove_spend_batch.argtypes = [vp, sz, i32, u8p, u8p, u8p, u8p, u8p, u8p]
    lib.act_prove_spend_seeded_batch.argtypes = [vp, sz, i32, u8p, u8p, u8p, C.c_uint64, u8p, u8p, u8p]
    lib.act_node_prove_spend_seeded_batch.argtypes = [vp, sz, u8p, u8p, u8p, C.c_uint64, u8p, u8p, u8p]
    lib.act_verify_spend_batch.argtypes = [vp, sz, i32, u8p, u8p, u8p, u8p]
    lib.act_refund_batch.argtypes = [vp, sz, i32, u8p, u8p, u8p, i32, u8p, u8p]
    lib.act_refund_to_credit_token_batch.argtypes = [vp, sz, i32, u8p, u8p, u8p, u8p, u8p, u8p]
    lib.act_debug_last_spend_transcripts.argtypes = [vp, sz, u8p, C.POINTER(sz)]
    lib.act_debug_scalarmult_batch.argtypes = [vp, sz, i32, u8p, u8p, u8p, u8p]
    lib.act_debug_secret_residue.argtypes = [vp, C.POINTER(sz)]
    for f in ("act_cbor_size", "act_cbor_record_bytes"):
        getattr(lib, f).argtypes = [vp, i32]
        getattr(lib, f).restype = sz
    lib.act_cbor_encode_batch.argtypes = [vp, i32, sz, i32, u8p, u8p]
    lib.act_cbor_decode_batch.argtypes = [vp, i32, sz, i32, u8p, vp, u8p, u8p]
    lib.act_verify_spend_cbor_batch.argtypes = [vp, sz, i32, u8p, u8p, vp, u8p, u8p]
    lib.act_node_verify_spend_cbor_batch.argtypes = [vp, sz, u8p, u8p, vp, u8p, u8p]
    lib.act_redeem_batch.argtypes = [vp, vp, sz, i32, u8p, u8p, u8p, i32, u8p, u8p]
    lib.act_node_redeem_batch.argtypes = [vp, vp, sz, u8p, u8p, u8p, i32, u8p, u8p]
    lib.act_nullifier_set_create.argtypes = [i32, sz, u8p, C.POINTER(vp)]
    lib.act_nullifier_set_destroy.argtypes = [vp]
    lib.act_nullifier_set_destroy.restype = None
    lib.act_nullifier_set_len.argtypes = [vp]
    lib.act_nullifier_set_len.restype = sz
    lib.act_nullifier_set_last_error.argtypes = [vp]
    lib.act_nullifier_set_last_error.restype = C.c_char_p
    lib.act_nullifier_check_and_insert_batch.argtypes = [vp, sz, i32, u8p, sz, u8p, u8p]
    lib.act_prof_enable.argtypes = [vp, i32]
    lib.act_prof_reset.argtypes = [vp]
    lib.act_prof_kernel_count.argtypes = [vp]
    lib.act_prof_kernel_name.argtypes = [vp, i32]
    lib.act_prof_kernel_name.restype = C.c_char_p
    lib.act_prof_get.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.act_prof_get_busy.argtypes = [vp, i32, C.POINTER(C.c_double)]
    lib.act_ubench_mad_u64_u32.argtypes = [i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.act_ubench_random_read.argtypes = [i32, sz, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.act_ubench_table_read.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.act_issue_check_batch.argtypes = [vp, sz, i32, u8p, u8p]
    lib.act_issue_sign_batch.argtypes = [vp, sz, i32, u8p, u8p, u8p, u8p, u8p, i32, u8p, u8p]
    lib.act_refund_sign_batch.argtypes = [vp, sz, i32, u8p, u8p, u8p, u8p, i32, u8p, u8p]
    lib.act_node_create.argtypes = [u8p, i32, C.POINTER(C.c_int), i32, sz, C.POINTER(vp)]
    lib.act_node_destroy.argtypes = [vp]
    lib.act_node_destroy.restype = None
    lib.act_node_device_count.argtypes = [vp]
    lib.act_node_ctx.argtypes = [vp, i32]
    lib.act_node_ctx.restype = vp
    lib.act_node_last_error.argtypes = [vp]
    lib.act_node_last_error.restype = C.c_char_p
    lib.act_node_set_transcript_mode.argtypes = [vp, i32]
    lib.act_node_set_host_threads.argtypes = [vp, i32]
    lib.act_node_request_batch.argtypes = [vp, sz, u8p, u8p, u8p]
    lib.act_node_issue_batch.argtypes = [vp, sz, u8p, u8p, u8p, u8p, i32, u8p, u8p]
    lib.act_node_issuance_to_credit_token_batch.argtypes = [vp, sz, u8p, u8p, u8p, u8p, u8p, u8p]
    lib.act_node_prove_spend_batch.argtypes = [vp, sz, u8p, u8p, u8p, u8p, u8p, u8p]
    lib.act_node_verify_spend_batch.argtypes = [vp, sz, u8p, u8p, u8p, u8p]
    lib.act_node_refund_batch.argtypes = [vp, sz, u8p, u8p, u8p, i32, u8p, u8p]
    lib.act_node_refund_to_credit_token_batch.argtypes = [vp, sz, u8p, u8p, u8p, u8p, u8p, u8p]
    lib.act_node_issue_check_batch.argtypes = [vp, sz, u8p, u8p]
    lib.act_node_issue_sign_batch.argtypes = [vp, sz, u8p, u8p, u8p, u8p, u8p, i32, u8p, u8p]
    lib.act_node_refund_sign_batch.argtypes = [vp, sz, u8p, u8p, u8p, u8p, i32, u8p, u8p]
    lib.act_node_nullifier_set_create.argtypes = [C.POINTER(C.c_int), i32, sz, u8p, C.POINTER(vp)]
    lib.act_node_nullifier_set_destroy.argtypes = [vp]
    lib.act_node_nullifier_set_destroy.restype = None
    lib.act_node_nullifier_set_len.argtypes = [vp]
    lib.act_node_nullifier_set_len.restype = sz
    lib.act_node_nullifier_set_last_error.argtypes = [vp]
    lib.act_node_nullifier_set_last_error.restype = C.c_char_p
    lib.act_node_nullifier_check_and_insert_batch.argtypes = [vp, sz, u8p, sz, u8p, u8p]
    lib.act_verify_spend_cbor_keys_batch.argtypes = [vp, sz, i32, u8p, u8p, vp, u8p, u8p, u8p]
    lib.act_node_verify_spend_cbor_keys_batch.argtypes = [vp, sz, u8p, u8p, vp, u8p, u8p, u8p]
    lib.act_refund_sign_cbor_batch.argtypes = [vp, sz, i32, u8p, u8p, u8p, u8p, i32, u8p, u8p]
    lib.act_refund_cbor_batch.argtypes = [vp, sz, i32, u8p, u8p, vp, u8p, i32, u8p, u8p]
    lib.act_refund_cbor_keys_batch.argtypes = [vp, sz, i32, u8p, u8p, vp, u8p, i32, u8p, u8p, u8p]
    lib.act_node_refund_sign_cbor_batch.argtypes = [vp, sz, u8p, u8p, u8p, u8p, i32, u8p, u8p]
    lib.act_node_refund_cbor_batch.argtypes = [vp, sz, u8p, u8p, vp, u8p, i32, u8p, u8p]
    lib.act_redeem_cbor_batch.argtypes = [vp, vp, sz, i32, u8p, u8p, vp, u8p, i32, u8p, u8p]
    lib.act_node_redeem_cbor_batch.argtypes = [vp, vp, sz, u8p, u8p, vp, u8p, i32, u8p, u8p]
    lib.act_ctx_host_hash_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64), i32]
    lib.act_ctx_set_tiny_calls.argtypes = [vp, i32]
    lib.act_node_set_balance.argtypes = [vp, i32, i32]
    lib.act_node_device_stats.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.act_node_balance_state.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.act_debug_set_slowdown.argtypes = [vp, C.c_uint32]
    lib.act_debug_fail_next_signs.argtypes = [vp, i32]
    _lib = lib
    forward_tuning_env(lib)
    return lib


# The library reads no ACT_* tuning variable itself (act_tuning_set is the only way in): tools/, tests/ and bench.py keep their
# environment-variable interface HERE, in the binding they all go through.  ACT_<NAME>=v -> act_tuning_set("<name>", v).
TUNING_ENV = ("NO_MAPPED_READS", "NO_STREAM_PROBE", "NO_FUSED_TINY", "NO_TAPER", "NO_WIDE_CLIENT", "NO_WIDE_PROVE", "NO_WIDE_SIGN", "NO_LDS_ISOLATION",
              "SMALL_NORMAL_PRIO", "SMALL_TRACE", "SMALL_IN_FLIGHT", "SMALL_SUB", "STAGGER", "HARD_STAGGER", "HOST_CHUNK", "CBOR_CHUNK_MSGS", "UBENCH_ITERS")


def forward_tuning_env(lib=None):
    """(also callable later: a test that changes ACT_CBOR_CHUNK_MSGS between calls re-forwards)"""
    lib = lib or load()
    defaults = {"STAGGER": -1, "SMALL_IN_FLIGHT": 2}
    for name in TUNING_ENV:
        v = os.environ.get("ACT_" + name)
        try:
            val = defaults.get(name, 0) if v is None else (int(v) if v.strip() else 1)
        except ValueError:
            val = 1
        lib.act_tuning_set(name.lower().encode(), val)


RNG_DRAW_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)


def rng_trampoline(fill):
    """act_rng_draw_fn around `fill(n) -> n bytes`.  ctypes swallows an exception raised inside a callback (it prints it and returns
    garbage), so the trampoline catches everything itself, checks the length, and reports failure as the ABI says: non-zero -- the
    library then signs nothing (ACT_ERR_RNG).  The exception is kept in `.error` of the returned callback's holder list."""
    err = []

    def draw(_ctx, dst, n):
        try:
            data = fill(n)
            if len(data) != n:
                raise ValueError("generator returned %d bytes, %d asked" % (len(data), n))
            C.memmove(dst, bytes(data), n)
            return 0
        except BaseException as e:      # noqa: BLE001 -- nothing may propagate into C
            err.append(e)
            return 1
    cb = RNG_DRAW_FN(draw)
    cb.errors = err
    return cb


class RngSource(C.Structure):
    """act_rng_source (ACT_RNG_CALLBACK): the library draws 128 bytes per SIGNED lane through `draw`, once, after the verdicts."""
    _fields_ = [("draw", RNG_DRAW_FN), ("rng_ctx", C.c_void_p)]


class ReplayRng:
    """A generator that hands out a fixed byte string (what tests/parity.rs feeds the crate); `pos` = bytes consumed so far."""

    def __init__(self, data: bytes):
        self.data, self.pos, self.draws = bytes(data), 0, []

        def fill(n):
            if self.pos + n > len(self.data):
                raise ValueError("generator exhausted")
            out = self.data[self.pos:self.pos + n]
            self.pos += n
            self.draws.append(n)
            return out
        self._cb = rng_trampoline(fill)
        self.source = RngSource(self._cb, None)

    @property
    def ptr(self):
        return C.addressof(self.source)


def _msgs(messages):
    """list of byte strings -> (blob pointer, keep-alive, offsets array)"""
    n = len(messages)
    offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([len(m) for m in messages], dtype=np.uint64)
    p, keep = _in(b"".join(messages) + b"\0")
    return p, keep, offs


def _rng_arg(rng):
    """bytes -> pointer; ReplayRng -> pointer to its act_rng_source (rng_mode must then be RNG_CALLBACK)"""
    if isinstance(rng, ReplayRng):
        return rng.ptr, rng
    return _in(rng)


def _in(x, nbytes=None):
    """bytes / numpy uint8 -> (ctypes pointer value, keep-alive object)."""
    if x is None:
        return None, None
    a = np.frombuffer(x, dtype=np.uint8) if isinstance(x, (bytes, bytearray, memoryview)) else np.ascontiguousarray(x, dtype=np.uint8)
    if nbytes is not None and a.size != nbytes:
        raise ValueError(f"expected {nbytes} bytes, got {a.size}")
    return a.ctypes.data, a


def params_new(org: str, svc: str, dep: str, ver: str, device: int = 0) -> bytes:
    out = np.zeros(96, np.uint8)
    rc = load().act_params_new(device, org.encode(), svc.encode(), dep.encode(), ver.encode(), out.ctypes.data)
    if rc:
        raise ActError(f"act_params_new failed: {_ERRS.get(rc, rc)}")
    return out.tobytes()


def params_random(rng: bytes, device: int = 0) -> bytes:
    out = np.zeros(96, np.uint8)
    p, keep = _in(rng, 192)
    rc = load().act_params_random(device, p, out.ctypes.data)
    if rc:
        raise ActError(f"act_params_random failed: {_ERRS.get(rc, rc)}")
    return out.tobytes()


def host_usable_cpus() -> int:
    return load().act_host_usable_cpus()


def host_pool_stats() -> dict:
    j, t, n = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
    load().act_host_pool_stats(C.byref(j), C.byref(t), C.byref(n))
    return {"jobs": j.value, "threads_created": t.value, "pool_size": n.value}


def host_hash_many(msgs, stride: int, length: int, n: int, max_threads: int = 0) -> bytes:
    """BLAKE3 XOF (64 B) of n messages at `stride` through the process-wide worker pool (pure host code)."""
    out = np.zeros(64 * n, np.uint8)
    p, keep = _in(msgs)
    load().act_host_hash_many(p, stride, length, n, max_threads, out.ctypes.data)
    return out.tobytes()


def ubench_mad(device: int = 0):
    """(lane-MADs per second, probe ms) of the v_mad_u64_u32 roofline probe."""
    r, ms = C.c_double(0), C.c_double(0)
    rc = load().act_ubench_mad_u64_u32(device, C.byref(r), C.byref(ms))
    if rc:
        raise ActError(f"act_ubench_mad_u64_u32 failed: {_ERRS.get(rc, rc)}")
    return r.value, ms.value


def ubench_random_read(device: int = 0, gib: int = 0, waves_per_simd: int = 0, in_flight: int = 0):
    """(GB/s of 128-byte random reads, probe ms): the memory-side roofline of the scalar-addressed fixed-base tables."""
    r, ms = C.c_double(0), C.c_double(0)
    rc = load().act_ubench_random_read(device, gib, waves_per_simd, in_flight, C.byref(r), C.byref(ms))
    if rc:
        raise ActError(f"act_ubench_random_read failed: {_ERRS.get(rc, rc)}")
    return r.value, ms.value


class Engine:
    """One context: Params (enc(h1)|enc(h2)|enc(h3)) + range width L + one GPU.
    Host-memory methods take/return bytes; the *_dev methods take raw device pointers (ints)."""

    def __init__(self, h: bytes, L: int = 128, device: int = 0, max_batch: int = 0,
                 transcript: int = TRANSCRIPT_HOST, host_threads: int = 0):
        self.lib = load()
        self.h, self.L, self.device = bytes(h), L, device
        ctx = C.c_void_p()
        p, keep = _in(h, 96)
        rc = self.lib.act_ctx_create(p, L, device, max_batch, C.byref(ctx))
        if rc:
            msg = self.lib.act_last_error(ctx).decode() if ctx else ""
            if ctx:
                self.lib.act_ctx_destroy(ctx)
            raise ActError(f"act_ctx_create failed: {_ERRS.get(rc, rc)} {msg}")
        self.ctx = ctx
        forward_tuning_env(self.lib)       # ACT_* measurement knobs as they stand now (tests change them between engines)
        wide = os.environ.get("ACT_FB_WIDE_BITS")          # tools' A/B interface of rounds 3-5; the library itself no longer looks
        if wide and wide.isdigit() and int(wide) != 16:
            for b in ((0, 1, 2, 3) if os.environ.get("ACT_FB_ALL_WIDE") else (1, 3)):
                self.lib.act_ctx_set_fixed_base_bits(ctx, b, int(wide))
        self.proof_bytes = self.lib.act_spend_proof_bytes(ctx)
        self.prove_rng_bytes = self.lib.act_prove_rng_bytes(ctx)
        self.transcript_bytes = self.lib.act_spend_transcript_bytes(ctx)
        self.set_transcript_mode(transcript)
        if host_threads:
            self._ck(self.lib.act_ctx_set_host_threads(ctx, host_threads))

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.act_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise ActError(f"{_ERRS.get(rc, rc)}: {self.lib.act_last_error(self.ctx).decode()}")

    def set_transcript_mode(self, mode: int):
        self._ck(self.lib.act_ctx_set_transcript_mode(self.ctx, mode))

    def fixed_base_bits(self):
        """Window widths of the g, h1, h2, h3 tables of this context."""
        return [self.lib.act_ctx_fixed_base_bits(self.ctx, b) for b in range(4)]

    def set_pipeline_depth(self, depth: int):
        self._ck(self.lib.act_ctx_set_pipeline_depth(self.ctx, depth))

    def set_host_threads(self, n: int):
        self._ck(self.lib.act_ctx_set_host_threads(self.ctx, n))

    def ubench_table_read(self, base: int, waves_per_simd: int = 0, in_flight: int = 0):
        """(GB/s, ms) of random 128-byte reads over this context's own table of base 0..3 (g, h1, h2, h3)."""
        r, ms = C.c_double(0), C.c_double(0)
        self._ck(self.lib.act_ubench_table_read(self.ctx, base, waves_per_simd, in_flight, C.byref(r), C.byref(ms)))
        return r.value, ms.value

    def set_fixed_base_bits(self, base: int, bits: int):
        """Window width of the table of base 0..3 (g, h1, h2, h3); act_ctx_create leaves all four at 16 bits."""
        self._ck(self.lib.act_ctx_set_fixed_base_bits(self.ctx, base, bits))

    def set_wide_range_tables(self, bits: int = 24) -> bool:
        """24-bit windows for h1 and h3, the range kernel's bases (+47 GB, +3 % verifies/s): what a GPU that serves nothing else asks
        for.  Returns False (tables unchanged) when the device has not the room."""
        try:
            self.set_fixed_base_bits(1, bits); self.set_fixed_base_bits(3, bits)
            return True
        except ActError:
            for b in (1, 3):
                if self.lib.act_ctx_fixed_base_bits(self.ctx, b) != 16:
                    self.lib.act_ctx_set_fixed_base_bits(self.ctx, b, 16)
            return False

    def set_small_batch_max(self, n: int):
        """Calls of at most n proofs take the small-batch (latency) schedule; 0 = never."""
        self._ck(self.lib.act_ctx_set_small_batch_max(self.ctx, n))

    def streams_overlap(self) -> int:
        """1 = the two pipeline streams run side by side, 0 = they share a hardware queue, -1 = not measured."""
        return self.lib.act_ctx_streams_overlap(self.ctx)

    # ---- host-memory batch calls ----------------------------------------------------------------
    def private_key_random(self, rng: bytes) -> bytes:
        out = np.zeros(64, np.uint8); p, k = _in(rng, 64)
        self._ck(self.lib.act_private_key_random(self.ctx, p, out.ctypes.data)); return out.tobytes()

    def pre_issuance_random(self, rng: bytes) -> bytes:
        n = len(rng) // 128; out = np.zeros(64 * n, np.uint8); p, k = _in(rng, 128 * n)
        self._ck(self.lib.act_pre_issuance_random_batch(self.ctx, n, MEM_HOST, p, out.ctypes.data)); return out.tobytes()

    def request(self, pre: bytes, rng: bytes) -> bytes:
        n = len(pre) // 64; out = np.zeros(128 * n, np.uint8)
        p0, k0 = _in(pre, 64 * n); p1, k1 = _in(rng, 128 * n)
        self._ck(self.lib.act_request_batch(self.ctx, n, MEM_HOST, p0, p1, out.ctypes.data)); return out.tobytes()

    def issue(self, sk: bytes, req: bytes, c: bytes, rng: bytes, rng_mode: int = RNG_PER_LANE):
        n = len(req) // 128; out = np.zeros(160 * n, np.uint8); st = np.zeros(n, np.uint8)
        ps, ks = _in(sk, 64); p0, k0 = _in(req, 128 * n); p1, k1 = _in(c, 32 * n); p2, k2 = _in(rng)
        self._ck(self.lib.act_issue_batch(self.ctx, n, MEM_HOST, ps, p0, p1, p2, rng_mode, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def issuance_to_credit_token(self, pre: bytes, w: bytes, req: bytes, resp: bytes):
        n = len(pre) // 64; out = np.zeros(160 * n, np.uint8); st = np.zeros(n, np.uint8)
        p0, k0 = _in(pre, 64 * n); pw, kw = _in(w, 32); p1, k1 = _in(req, 128 * n); p2, k2 = _in(resp, 160 * n)
        self._ck(self.lib.act_issuance_to_credit_token_batch(self.ctx, n, MEM_HOST, p0, pw, p1, p2, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def prove_spend(self, tok: bytes, s: bytes, rng: bytes):
        n = len(tok) // 160; out = np.zeros(self.proof_bytes * n, np.uint8); pr = np.zeros(96 * n, np.uint8); st = np.zeros(n, np.uint8)
        p0, k0 = _in(tok, 160 * n); p1, k1 = _in(s, 32 * n); p2, k2 = _in(rng, self.prove_rng_bytes * n)
        self._ck(self.lib.act_prove_spend_batch(self.ctx, n, MEM_HOST, p0, p1, p2, out.ctypes.data, pr.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes(), pr.tobytes()

    def prove_spend_seeded(self, tok: bytes, s: bytes, seed: bytes, first_lane: int = 0):
        """prove_spend with lane i's rng = BLAKE3-XOF(seed | u64_le(first_lane + i)), expanded on the device."""
        n = len(tok) // 160; out = np.zeros(self.proof_bytes * n, np.uint8); pr = np.zeros(96 * n, np.uint8); st = np.zeros(n, np.uint8)
        p0, k0 = _in(tok, 160 * n); p1, k1 = _in(s, 32 * n); p2, k2 = _in(seed, 32)
        self._ck(self.lib.act_prove_spend_seeded_batch(self.ctx, n, MEM_HOST, p0, p1, p2, first_lane, out.ctypes.data, pr.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes(), pr.tobytes()

    def verify_spend(self, sk: bytes, proofs: bytes, want_kprime: bool = False):
        n = len(proofs) // self.proof_bytes; st = np.zeros(n, np.uint8)
        kp = np.zeros(32 * n, np.uint8) if want_kprime else None
        ps, ks = _in(sk, 64); p0, k0 = _in(proofs, self.proof_bytes * n)
        self._ck(self.lib.act_verify_spend_batch(self.ctx, n, MEM_HOST, ps, p0, st.ctypes.data, kp.ctypes.data if want_kprime else None))
        return (st.tobytes(), kp.tobytes()) if want_kprime else st.tobytes()

    def refund(self, sk: bytes, proofs: bytes, rng: bytes, rng_mode: int = RNG_PER_LANE):
        n = len(proofs) // self.proof_bytes; out = np.zeros(128 * n, np.uint8); st = np.zeros(n, np.uint8)
        ps, ks = _in(sk, 64); p0, k0 = _in(proofs, self.proof_bytes * n); p1, k1 = _in(rng)
        self._ck(self.lib.act_refund_batch(self.ctx, n, MEM_HOST, ps, p0, p1, rng_mode, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def refund_to_credit_token(self, prerefund: bytes, proofs: bytes, refund: bytes, w: bytes):
        n = len(prerefund) // 96; out = np.zeros(160 * n, np.uint8); st = np.zeros(n, np.uint8)
        p0, k0 = _in(prerefund, 96 * n); p1, k1 = _in(proofs, self.proof_bytes * n); p2, k2 = _in(refund, 128 * n); pw, kw = _in(w, 32)
        self._ck(self.lib.act_refund_to_credit_token_batch(self.ctx, n, MEM_HOST, p0, p1, p2, pw, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def last_spend_transcripts(self, max_lanes: int) -> list:
        out = np.zeros(max_lanes * self.transcript_bytes, np.uint8); n = C.c_size_t(0)
        self._ck(self.lib.act_debug_last_spend_transcripts(self.ctx, max_lanes, out.ctypes.data, C.byref(n)))
        b = out.tobytes()
        return [b[i * self.transcript_bytes:(i + 1) * self.transcript_bytes] for i in range(n.value)]

    def secret_residue(self) -> int:
        n = C.c_size_t(0)
        self._ck(self.lib.act_debug_secret_residue(self.ctx, C.byref(n))); return n.value

    def debug_scalarmult(self, points: bytes, scalars: bytes):
        n = len(points) // 32; out = np.zeros(32 * n, np.uint8); st = np.zeros(n, np.uint8)
        p0, k0 = _in(points, 32 * n); p1, k1 = _in(scalars, 32 * n)
        self._ck(self.lib.act_debug_scalarmult_batch(self.ctx, n, MEM_HOST, p0, p1, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    # ---- CBOR wire codec (src/cbor.rs) -----------------------------------------------------------------
    def cbor_size(self, type_name: str) -> int:
        return self.lib.act_cbor_size(self.ctx, CBOR_TYPES[type_name])

    def cbor_encode(self, type_name: str, records: bytes) -> list:
        t = CBOR_TYPES[type_name]; rb = self.lib.act_cbor_record_bytes(self.ctx, t); ml = self.lib.act_cbor_size(self.ctx, t)
        n = len(records) // rb; out = np.zeros(ml * n, np.uint8); p0, k0 = _in(records, rb * n)
        self._ck(self.lib.act_cbor_encode_batch(self.ctx, t, n, MEM_HOST, p0, out.ctypes.data))
        b = out.tobytes()
        return [b[i * ml:(i + 1) * ml] for i in range(n)]

    def cbor_decode(self, type_name: str, messages: list):
        """messages: list of byte strings (any lengths).  Returns (status bytes, records bytes)."""
        t = CBOR_TYPES[type_name]; rb = self.lib.act_cbor_record_bytes(self.ctx, t); n = len(messages)
        offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([len(m) for m in messages], dtype=np.uint64)
        blob = b"".join(messages) + b"\0"
        out = np.zeros(rb * n, np.uint8); st = np.zeros(n, np.uint8); p0, k0 = _in(blob)
        self._ck(self.lib.act_cbor_decode_batch(self.ctx, t, n, MEM_HOST, p0, offs.ctypes.data, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def redeem(self, nullifier_set, sk: bytes, proofs: bytes, rng: bytes, rng_mode: int = RNG_PER_LANE):
        """verify -> nullifier check-and-insert -> sign: statuses (3 = DoubleSpendError) and refunds."""
        n = len(proofs) // self.proof_bytes; out = np.zeros(128 * n, np.uint8); st = np.zeros(n, np.uint8)
        ps, ks = _in(sk, 64); p0, k0 = _in(proofs, self.proof_bytes * n); p1, k1 = _rng_arg(rng)
        self._ck(self.lib.act_redeem_batch(self.ctx, nullifier_set.h, n, MEM_HOST, ps, p0, p1, rng_mode, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    # the halves of issue / refund around the point where the crate draws its rng (src/lib.rs:643, 846): check, then sign what was accepted
    def issue_check(self, req: bytes) -> bytes:
        n = len(req) // 128; st = np.zeros(n, np.uint8); p0, k0 = _in(req, 128 * n)
        self._ck(self.lib.act_issue_check_batch(self.ctx, n, MEM_HOST, p0, st.ctypes.data)); return st.tobytes()

    def issue_sign(self, sk: bytes, req: bytes, c: bytes, status_in: bytes, rng: bytes, rng_mode: int = RNG_SEQUENTIAL):
        n = len(status_in); out = np.zeros(160 * n, np.uint8); st = np.zeros(n, np.uint8)
        ps, ks = _in(sk, 64); p0, k0 = _in(req, 128 * n); p1, k1 = _in(c, 32 * n); p2, k2 = _in(status_in, n); p3, k3 = _in(rng)
        self._ck(self.lib.act_issue_sign_batch(self.ctx, n, MEM_HOST, ps, p0, p1, p2, p3, rng_mode, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def refund_sign(self, sk: bytes, kprime: bytes, status_in: bytes, rng: bytes, rng_mode: int = RNG_SEQUENTIAL):
        n = len(status_in); out = np.zeros(128 * n, np.uint8); st = np.zeros(n, np.uint8)
        ps, ks = _in(sk, 64); p0, k0 = _in(kprime, 32 * n); p1, k1 = _in(status_in, n); p2, k2 = _in(rng)
        self._ck(self.lib.act_refund_sign_batch(self.ctx, n, MEM_HOST, ps, p0, p1, p2, rng_mode, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def redeem_dev(self, nullifier_set, sk: bytes, n: int, d_proofs: int, d_rng: int, rng_mode: int, d_out: int, d_status: int):
        ps, ks = _in(sk, 64)
        self._ck(self.lib.act_redeem_batch(self.ctx, nullifier_set.h, n, MEM_DEVICE, ps, d_proofs, d_rng, rng_mode, d_out, d_status))

    def verify_spend_cbor(self, sk: bytes, messages: list, want_kprime: bool = False):
        """CBOR SpendProof messages (byte strings of any length) -> statuses: from_cbor + refund's verification in one pass."""
        n = len(messages)
        offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([len(m) for m in messages], dtype=np.uint64)
        blob = b"".join(messages) + b"\0"
        st = np.zeros(n, np.uint8); kp = np.zeros(32 * n, np.uint8) if want_kprime else None
        ps, ks = _in(sk, 64); p0, k0 = _in(blob)
        self._ck(self.lib.act_verify_spend_cbor_batch(self.ctx, n, MEM_HOST, ps, p0, offs.ctypes.data, st.ctypes.data, kp.ctypes.data if want_kprime else None))
        return (st.tobytes(), kp.tobytes()) if want_kprime else st.tobytes()

    def verify_spend_cbor_ptr(self, sk: bytes, n: int, mem: int, p_cbor: int, p_offsets: int, p_status: int, p_kprime: int = 0):
        ps, ks = _in(sk, 64)
        self._ck(self.lib.act_verify_spend_cbor_batch(self.ctx, n, mem, ps, p_cbor, p_offsets or None, p_status, p_kprime or None))

    def verify_spend_cbor_keys(self, sk: bytes, messages: list):
        """-> (statuses, enc(K') per message, nullifier `k` per message as it stood on the wire)"""
        n = len(messages); p0, k0, offs = _msgs(messages)
        st = np.zeros(n, np.uint8); kp = np.zeros(32 * n, np.uint8); nul = np.zeros(32 * n, np.uint8); ps, ks = _in(sk, 64)
        self._ck(self.lib.act_verify_spend_cbor_keys_batch(self.ctx, n, MEM_HOST, ps, p0, offs.ctypes.data, st.ctypes.data, kp.ctypes.data, nul.ctypes.data))
        return st.tobytes(), kp.tobytes(), nul.tobytes()

    def refund_cbor(self, sk: bytes, messages: list, rng, rng_mode: int = RNG_PER_LANE):
        """CBOR SpendProof messages in -> (statuses, list of CBOR Refund messages; b"" for a lane that was not signed)"""
        n = len(messages); p0, k0, offs = _msgs(messages); ml = self.cbor_size("Refund")
        st = np.zeros(n, np.uint8); out = np.zeros(ml * n, np.uint8); ps, ks = _in(sk, 64); pr, kr = _rng_arg(rng)
        self._ck(self.lib.act_refund_cbor_batch(self.ctx, n, MEM_HOST, ps, p0, offs.ctypes.data, pr, rng_mode, out.ctypes.data, st.ctypes.data))
        b = out.tobytes()
        assert all(st[i] == 0 or not out[i * ml:(i + 1) * ml].any() for i in range(n)), "a failed lane's slot is not zero"
        return st.tobytes(), [b[i * ml:(i + 1) * ml] if st[i] == 0 else b"" for i in range(n)]

    def refund_sign_cbor(self, sk: bytes, kprime: bytes, status_in: bytes, rng, rng_mode: int = RNG_SEQUENTIAL):
        n = len(status_in); ml = self.cbor_size("Refund"); st = np.zeros(n, np.uint8); out = np.zeros(ml * n, np.uint8)
        ps, ks = _in(sk, 64); p0, k0 = _in(kprime, 32 * n); p1, k1 = _in(status_in, n); pr, kr = _rng_arg(rng)
        self._ck(self.lib.act_refund_sign_cbor_batch(self.ctx, n, MEM_HOST, ps, p0, p1, pr, rng_mode, out.ctypes.data, st.ctypes.data))
        b = out.tobytes()
        return st.tobytes(), [b[i * ml:(i + 1) * ml] if st[i] == 0 else b"" for i in range(n)]

    def redeem_cbor(self, nullifier_set, sk: bytes, messages: list, rng, rng_mode: int = RNG_SEQUENTIAL, raw: bool = False):
        """wire bytes -> verify -> nullifier check-and-insert -> sign -> wire bytes.  raw=True: (rc, statuses, out bytes), no exception."""
        n = len(messages); p0, k0, offs = _msgs(messages); ml = self.cbor_size("Refund")
        st = np.zeros(n, np.uint8); out = np.full(ml * n, 7 if raw else 0, np.uint8); ps, ks = _in(sk, 64); pr, kr = _rng_arg(rng)
        rc = self.lib.act_redeem_cbor_batch(self.ctx, nullifier_set.h, n, MEM_HOST, ps, p0, offs.ctypes.data, pr, rng_mode, out.ctypes.data, st.ctypes.data)
        if raw:
            return rc, st.tobytes(), out.tobytes()
        self._ck(rc)
        b = out.tobytes()
        return st.tobytes(), [b[i * ml:(i + 1) * ml] if st[i] == 0 else b"" for i in range(n)]

    def wire_ptr(self, fn: str, sk: bytes, n: int, mem: int, p_cbor: int, p_offsets: int, p_rng: int, rng_mode: int, p_out: int, p_status: int, nullifier_set=None):
        """act_refund_cbor_batch / act_redeem_cbor_batch on raw pointers of either kind (device-memory callers)"""
        ps, ks = _in(sk, 64)
        if fn == "refund":
            self._ck(self.lib.act_refund_cbor_batch(self.ctx, n, mem, ps, p_cbor, p_offsets or None, p_rng, rng_mode, p_out, p_status))
        else:
            self._ck(self.lib.act_redeem_cbor_batch(self.ctx, nullifier_set.h, n, mem, ps, p_cbor, p_offsets or None, p_rng, rng_mode, p_out, p_status))

    def host_hash_stats(self, reset: bool = False) -> dict:
        """host-transcript mode: seconds the calling thread waited for transcripts / hashed them, bytes hashed, since the last reset"""
        w, hs, b = C.c_double(0), C.c_double(0), C.c_uint64(0)
        self._ck(self.lib.act_ctx_host_hash_stats(self.ctx, C.byref(w), C.byref(hs), C.byref(b), 1 if reset else 0))
        return {"wait_s": w.value, "hash_s": hs.value, "bytes": b.value}

    def set_tiny_calls(self, on: bool):
        """calls of at most 64 lanes as one kernel with the transcript hashed in it (default) or the multi-launch paths"""
        self._ck(self.lib.act_ctx_set_tiny_calls(self.ctx, 1 if on else 0))

    def set_slowdown(self, ns_per_lane: int):
        self._ck(self.lib.act_debug_set_slowdown(self.ctx, ns_per_lane))

    # ---- device-memory batch calls (raw device pointers) -----------------------------------------
    def verify_spend_dev(self, sk: bytes, n: int, d_proofs: int, d_status: int, d_kprime: int = 0):
        ps, ks = _in(sk, 64)
        self._ck(self.lib.act_verify_spend_batch(self.ctx, n, MEM_DEVICE, ps, d_proofs, d_status, d_kprime or None))

    def verify_spend_ptr(self, sk: bytes, n: int, mem: int, p_proofs: int, p_status: int, p_kprime: int = 0):
        """Raw pointers of either kind (mem = MEM_HOST for e.g. pinned host buffers, MEM_DEVICE for HBM)."""
        ps, ks = _in(sk, 64)
        self._ck(self.lib.act_verify_spend_batch(self.ctx, n, mem, ps, p_proofs, p_status, p_kprime or None))

    def refund_dev(self, sk: bytes, n: int, d_proofs: int, d_rng: int, rng_mode: int, d_out: int, d_status: int):
        ps, ks = _in(sk, 64)
        self._ck(self.lib.act_refund_batch(self.ctx, n, MEM_DEVICE, ps, d_proofs, d_rng, rng_mode, d_out, d_status))

    def prove_spend_dev(self, n: int, d_tok: int, d_s: int, d_rng: int, d_proof: int, d_prerefund: int, d_status: int):
        self._ck(self.lib.act_prove_spend_batch(self.ctx, n, MEM_DEVICE, d_tok, d_s, d_rng, d_proof, d_prerefund, d_status))

    def issue_dev(self, sk: bytes, n: int, d_req: int, d_c: int, d_rng: int, rng_mode: int, d_out: int, d_status: int):
        ps, ks = _in(sk, 64)
        self._ck(self.lib.act_issue_batch(self.ctx, n, MEM_DEVICE, ps, d_req, d_c, d_rng, rng_mode, d_out, d_status))

    def request_dev(self, n: int, d_pre: int, d_rng: int, d_out: int):
        self._ck(self.lib.act_request_batch(self.ctx, n, MEM_DEVICE, d_pre, d_rng, d_out))

    def pre_issuance_random_dev(self, n: int, d_rng: int, d_out: int):
        self._ck(self.lib.act_pre_issuance_random_batch(self.ctx, n, MEM_DEVICE, d_rng, d_out))

    def issuance_to_credit_token_dev(self, n: int, d_pre: int, w: bytes, d_req: int, d_resp: int, d_tok: int, d_status: int):
        pw, kw = _in(w, 32)
        self._ck(self.lib.act_issuance_to_credit_token_batch(self.ctx, n, MEM_DEVICE, d_pre, pw, d_req, d_resp, d_tok, d_status))

    # ---- profiling ---------------------------------------------------------------------------------
    def prof_enable(self, on: bool = True):
        self._ck(self.lib.act_prof_enable(self.ctx, 1 if on else 0))

    def prof_reset(self):
        self._ck(self.lib.act_prof_reset(self.ctx))

    def prof(self) -> dict:
        out = {}
        for i in range(self.lib.act_prof_kernel_count(self.ctx)):
            ms, la, ln = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
            self._ck(self.lib.act_prof_get(self.ctx, i, C.byref(ms), C.byref(la), C.byref(ln)))
            if la.value:
                busy = C.c_double(0)
                self._ck(self.lib.act_prof_get_busy(self.ctx, i, C.byref(busy)))
                out[self.lib.act_prof_kernel_name(self.ctx, i).decode()] = {"ms": ms.value, "busy_ms": busy.value, "launches": la.value, "lanes": ln.value}
        return out


class Node:
    """The GPUs of one node behind one handle (act_node_*): contiguous shards, one context and host thread per entry of
    `devices`, outputs in the matching slices; ACT_RNG_SEQUENTIAL exact across shards.  Host memory (bytes) only."""

    def __init__(self, h: bytes, L: int = 128, devices=(0,), max_batch: int = 0, transcript: int = TRANSCRIPT_HOST):
        self.lib = load()
        self.L = L
        nd = C.c_void_p()
        p, keep = _in(h, 96)
        devs = (C.c_int * len(devices))(*devices)
        rc = self.lib.act_node_create(p, L, devs, len(devices), max_batch, C.byref(nd))
        if rc:
            msg = self.lib.act_node_last_error(nd).decode() if nd else ""
            if nd:
                self.lib.act_node_destroy(nd)
            raise ActError(f"act_node_create failed: {_ERRS.get(rc, rc)} {msg}")
        self.nd = nd
        c0 = self.lib.act_node_ctx(nd, 0)
        self.proof_bytes = self.lib.act_spend_proof_bytes(c0)
        self.prove_rng_bytes = self.lib.act_prove_rng_bytes(c0)
        self._ck(self.lib.act_node_set_transcript_mode(nd, transcript))

    def close(self):
        if getattr(self, "nd", None):
            self.lib.act_node_destroy(self.nd)
            self.nd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise ActError(f"{_ERRS.get(rc, rc)}: {self.lib.act_node_last_error(self.nd).decode()}")

    def device_count(self) -> int:
        return self.lib.act_node_device_count(self.nd)

    def set_transcript_mode(self, mode: int):
        self._ck(self.lib.act_node_set_transcript_mode(self.nd, mode))

    def set_host_threads(self, per_gpu: int):
        self._ck(self.lib.act_node_set_host_threads(self.nd, per_gpu))

    def set_fixed_base_bits(self, base: int, bits: int):
        self._ck(self.lib.act_node_set_fixed_base_bits(self.nd, base, bits))

    def streams_overlap(self) -> list:
        return [self.lib.act_ctx_streams_overlap(self.lib.act_node_ctx(self.nd, k)) for k in range(self.device_count())]

    def request(self, pre: bytes, rng: bytes) -> bytes:
        n = len(pre) // 64; out = np.zeros(128 * n, np.uint8)
        p0, k0 = _in(pre, 64 * n); p1, k1 = _in(rng, 128 * n)
        self._ck(self.lib.act_node_request_batch(self.nd, n, p0, p1, out.ctypes.data)); return out.tobytes()

    def issue(self, sk: bytes, req: bytes, c: bytes, rng: bytes, rng_mode: int = RNG_PER_LANE):
        n = len(req) // 128; out = np.zeros(160 * n, np.uint8); st = np.zeros(n, np.uint8)
        ps, ks = _in(sk, 64); p0, k0 = _in(req, 128 * n); p1, k1 = _in(c, 32 * n); p2, k2 = _in(rng)
        self._ck(self.lib.act_node_issue_batch(self.nd, n, ps, p0, p1, p2, rng_mode, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def issuance_to_credit_token(self, pre: bytes, w: bytes, req: bytes, resp: bytes):
        n = len(pre) // 64; out = np.zeros(160 * n, np.uint8); st = np.zeros(n, np.uint8)
        p0, k0 = _in(pre, 64 * n); pw, kw = _in(w, 32); p1, k1 = _in(req, 128 * n); p2, k2 = _in(resp, 160 * n)
        self._ck(self.lib.act_node_issuance_to_credit_token_batch(self.nd, n, p0, pw, p1, p2, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def prove_spend(self, tok: bytes, s: bytes, rng: bytes):
        n = len(tok) // 160; out = np.zeros(self.proof_bytes * n, np.uint8); pr = np.zeros(96 * n, np.uint8); st = np.zeros(n, np.uint8)
        p0, k0 = _in(tok, 160 * n); p1, k1 = _in(s, 32 * n); p2, k2 = _in(rng, self.prove_rng_bytes * n)
        self._ck(self.lib.act_node_prove_spend_batch(self.nd, n, p0, p1, p2, out.ctypes.data, pr.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes(), pr.tobytes()

    def prove_spend_seeded(self, tok: bytes, s: bytes, seed: bytes, first_lane: int = 0):
        n = len(tok) // 160; out = np.zeros(self.proof_bytes * n, np.uint8); pr = np.zeros(96 * n, np.uint8); st = np.zeros(n, np.uint8)
        p0, k0 = _in(tok, 160 * n); p1, k1 = _in(s, 32 * n); p2, k2 = _in(seed, 32)
        self._ck(self.lib.act_node_prove_spend_seeded_batch(self.nd, n, p0, p1, p2, first_lane, out.ctypes.data, pr.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes(), pr.tobytes()

    def verify_spend(self, sk: bytes, proofs: bytes, want_kprime: bool = False):
        n = len(proofs) // self.proof_bytes; st = np.zeros(n, np.uint8)
        kp = np.zeros(32 * n, np.uint8) if want_kprime else None
        ps, ks = _in(sk, 64); p0, k0 = _in(proofs, self.proof_bytes * n)
        self._ck(self.lib.act_node_verify_spend_batch(self.nd, n, ps, p0, st.ctypes.data, kp.ctypes.data if want_kprime else None))
        return (st.tobytes(), kp.tobytes()) if want_kprime else st.tobytes()

    def verify_spend_ptr(self, sk: bytes, n: int, p_proofs: int, p_status: int, p_kprime: int = 0):
        """Raw HOST pointers (pageable or pinned): the call a Rust caller's slices turn into."""
        ps, ks = _in(sk, 64)
        self._ck(self.lib.act_node_verify_spend_batch(self.nd, n, ps, p_proofs, p_status, p_kprime or None))

    def refund(self, sk: bytes, proofs: bytes, rng: bytes, rng_mode: int = RNG_PER_LANE):
        n = len(proofs) // self.proof_bytes; out = np.zeros(128 * n, np.uint8); st = np.zeros(n, np.uint8)
        ps, ks = _in(sk, 64); p0, k0 = _in(proofs, self.proof_bytes * n); p1, k1 = _in(rng)
        self._ck(self.lib.act_node_refund_batch(self.nd, n, ps, p0, p1, rng_mode, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def redeem(self, nullifier_set, sk: bytes, proofs: bytes, rng: bytes, rng_mode: int = RNG_SEQUENTIAL):
        n = len(proofs) // self.proof_bytes; out = np.zeros(128 * n, np.uint8); st = np.zeros(n, np.uint8)
        ps, ks = _in(sk, 64); p0, k0 = _in(proofs, self.proof_bytes * n); p1, k1 = _rng_arg(rng)
        self._ck(self.lib.act_node_redeem_batch(self.nd, nullifier_set.h, n, ps, p0, p1, rng_mode, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def refund_sign(self, sk: bytes, kprime: bytes, status_in: bytes, rng: bytes, rng_mode: int = RNG_SEQUENTIAL):
        n = len(status_in); out = np.zeros(128 * n, np.uint8); st = np.zeros(n, np.uint8)
        ps, ks = _in(sk, 64); p0, k0 = _in(kprime, 32 * n); p1, k1 = _in(status_in, n); p2, k2 = _in(rng)
        self._ck(self.lib.act_node_refund_sign_batch(self.nd, n, ps, p0, p1, p2, rng_mode, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def issue_check(self, req: bytes) -> bytes:
        n = len(req) // 128; st = np.zeros(n, np.uint8); p0, k0 = _in(req, 128 * n)
        self._ck(self.lib.act_node_issue_check_batch(self.nd, n, p0, st.ctypes.data)); return st.tobytes()

    def issue_sign(self, sk: bytes, req: bytes, c: bytes, status_in: bytes, rng: bytes, rng_mode: int = RNG_SEQUENTIAL):
        n = len(status_in); out = np.zeros(160 * n, np.uint8); st = np.zeros(n, np.uint8)
        ps, ks = _in(sk, 64); p0, k0 = _in(req, 128 * n); p1, k1 = _in(c, 32 * n); p2, k2 = _in(status_in, n); p3, k3 = _in(rng)
        self._ck(self.lib.act_node_issue_sign_batch(self.nd, n, ps, p0, p1, p2, p3, rng_mode, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()

    def refund_to_credit_token(self, prerefund: bytes, proofs: bytes, refund: bytes, w: bytes):
        n = len(prerefund) // 96; out = np.zeros(160 * n, np.uint8); st = np.zeros(n, np.uint8)
        p0, k0 = _in(prerefund, 96 * n); p1, k1 = _in(proofs, self.proof_bytes * n); p2, k2 = _in(refund, 128 * n); pw, kw = _in(w, 32)
        self._ck(self.lib.act_node_refund_to_credit_token_batch(self.nd, n, p0, p1, p2, pw, out.ctypes.data, st.ctypes.data))
        return st.tobytes(), out.tobytes()


    def _wire(self, messages):
        p0, k0, offs = _msgs(messages)
        return p0, k0, offs, self.lib.act_cbor_size(self.lib.act_node_ctx(self.nd, 0), CBOR_TYPES["Refund"])

    def verify_spend_cbor_keys(self, sk: bytes, messages: list):
        n = len(messages); p0, k0, offs = _msgs(messages)
        st = np.zeros(n, np.uint8); kp = np.zeros(32 * n, np.uint8); nul = np.zeros(32 * n, np.uint8); ps, ks = _in(sk, 64)
        self._ck(self.lib.act_node_verify_spend_cbor_keys_batch(self.nd, n, ps, p0, offs.ctypes.data, st.ctypes.data, kp.ctypes.data, nul.ctypes.data))
        return st.tobytes(), kp.tobytes(), nul.tobytes()

    def refund_cbor(self, sk: bytes, messages: list, rng, rng_mode: int = RNG_SEQUENTIAL):
        n = len(messages); p0, k0, offs, ml = self._wire(messages)
        st = np.zeros(n, np.uint8); out = np.zeros(ml * n, np.uint8); ps, ks = _in(sk, 64); pr, kr = _rng_arg(rng)
        self._ck(self.lib.act_node_refund_cbor_batch(self.nd, n, ps, p0, offs.ctypes.data, pr, rng_mode, out.ctypes.data, st.ctypes.data))
        b = out.tobytes()
        return st.tobytes(), [b[i * ml:(i + 1) * ml] if st[i] == 0 else b"" for i in range(n)]

    def refund_sign_cbor(self, sk: bytes, kprime: bytes, status_in: bytes, rng, rng_mode: int = RNG_SEQUENTIAL):
        n = len(status_in); ml = self.lib.act_cbor_size(self.lib.act_node_ctx(self.nd, 0), CBOR_TYPES["Refund"])
        st = np.zeros(n, np.uint8); out = np.zeros(ml * n, np.uint8)
        ps, ks = _in(sk, 64); p0, k0 = _in(kprime, 32 * n); p1, k1 = _in(status_in, n); pr, kr = _rng_arg(rng)
        self._ck(self.lib.act_node_refund_sign_cbor_batch(self.nd, n, ps, p0, p1, pr, rng_mode, out.ctypes.data, st.ctypes.data))
        b = out.tobytes()
        return st.tobytes(), [b[i * ml:(i + 1) * ml] if st[i] == 0 else b"" for i in range(n)]

    def redeem_cbor(self, nullifier_set, sk: bytes, messages: list, rng, rng_mode: int = RNG_SEQUENTIAL):
        n = len(messages); p0, k0, offs, ml = self._wire(messages)
        st = np.zeros(n, np.uint8); out = np.zeros(ml * n, np.uint8); ps, ks = _in(sk, 64); pr, kr = _rng_arg(rng)
        self._ck(self.lib.act_node_redeem_cbor_batch(self.nd, nullifier_set.h, n, ps, p0, offs.ctypes.data, pr, rng_mode, out.ctypes.data, st.ctypes.data))
        b = out.tobytes()
        return st.tobytes(), [b[i * ml:(i + 1) * ml] if st[i] == 0 else b"" for i in range(n)]

    def set_balance(self, weighted: bool = True, tail_64ths: int = -1):
        self._ck(self.lib.act_node_set_balance(self.nd, 1 if weighted else 0, tail_64ths))

    def device_stats(self) -> list:
        """per context: weight (relative speed, mean 1) and what the most recent cut call gave it"""
        out = []
        for k in range(self.device_count()):
            w, s = C.c_double(0), C.c_double(0); la, ca = C.c_uint64(0), C.c_uint64(0)
            self._ck(self.lib.act_node_device_stats(self.nd, k, C.byref(w), C.byref(la), C.byref(s), C.byref(ca)))
            out.append({"weight": w.value, "lanes": la.value, "seconds": s.value, "calls": ca.value})
        return out

    def balance_state(self) -> dict:
        sp, tf = C.c_double(0), C.c_double(0)
        self._ck(self.lib.act_node_balance_state(self.nd, C.byref(sp), C.byref(tf)))
        return {"spread": sp.value, "tail_fraction": tf.value}

    def ctx_handle(self, k: int):
        return self.lib.act_node_ctx(self.nd, k)


class NullifierSet:
    """GPU double-spend set with the sequential meaning of the reference tests' NullifierDb (src/tests.rs:29-50)."""

    def __init__(self, capacity: int, device: int = 0, salt: bytes = None):
        self.lib = load()
        h = C.c_void_p()
        p, keep = _in(salt, 16) if salt else (None, None)
        rc = self.lib.act_nullifier_set_create(device, capacity, p, C.byref(h))
        if rc:
            raise ActError(f"act_nullifier_set_create failed: {_ERRS.get(rc, rc)}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.act_nullifier_set_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.lib.act_nullifier_set_len(self.h)

    def _ck(self, rc):
        if rc:
            raise ActError(f"{_ERRS.get(rc, rc)}: {self.lib.act_nullifier_set_last_error(self.h).decode()}")

    def check_and_insert(self, nullifiers: bytes, stride: int = 32, skip_mask: bytes = None) -> bytes:
        n = (len(nullifiers) + stride - 32) // stride if nullifiers else 0
        out = np.zeros(n, np.uint8)
        p0, k0 = _in(nullifiers); pm, km = _in(skip_mask, n) if skip_mask is not None else (None, None)
        self._ck(self.lib.act_nullifier_check_and_insert_batch(self.h, n, MEM_HOST, p0, stride, pm, out.ctypes.data))
        return out.tobytes()

    def check_and_insert_dev(self, n: int, d_nullifiers: int, stride: int, d_skip_mask: int, d_out_spent: int):
        self._ck(self.lib.act_nullifier_check_and_insert_batch(self.h, n, MEM_DEVICE, d_nullifiers, stride, d_skip_mask or None, d_out_spent))


class NodeNullifierSet:
    """The double-spend set over the GPUs of a node (act_node_nullifier_*): host-side routing by owner, one set per device."""

    def __init__(self, capacity_per_device: int, devices=(0,), salt: bytes = None):
        self.lib = load()
        h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        p, keep = _in(salt, 16) if salt else (None, None)
        rc = self.lib.act_node_nullifier_set_create(devs, len(devices), capacity_per_device, p, C.byref(h))
        if rc:
            msg = self.lib.act_node_nullifier_set_last_error(h).decode() if h else ""
            if h:
                self.lib.act_node_nullifier_set_destroy(h)
            raise ActError(f"act_node_nullifier_set_create failed: {_ERRS.get(rc, rc)} {msg}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.act_node_nullifier_set_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.lib.act_node_nullifier_set_len(self.h)

    def check_and_insert(self, nullifiers: bytes, stride: int = 32, skip_mask: bytes = None) -> bytes:
        n = (len(nullifiers) + stride - 32) // stride if nullifiers else 0
        out = np.zeros(n, np.uint8)
        p0, k0 = _in(nullifiers); pm, km = _in(skip_mask, n) if skip_mask is not None else (None, None)
        rc = self.lib.act_node_nullifier_check_and_insert_batch(self.h, n, p0, stride, pm, out.ctypes.data)
        if rc:
            raise ActError(f"{_ERRS.get(rc, rc)}: {self.lib.act_node_nullifier_set_last_error(self.h).decode()}")
        return out.tobytes()
