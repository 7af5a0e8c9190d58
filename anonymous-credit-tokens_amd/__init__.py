"""act_amd — MI355X (gfx950) batch engine for the sigma-protocol hot path of anonymous-credit-tokens.

`capi` is the ctypes binding of the C ABI (include/act_mi355x.h); `api` mirrors the reference crate's
types and method names over it.  The HIP library is mandatory: importing `capi.load()` raises if
libact_mi355x.so is missing, and every call fails without a GPU — there is no CPU fallback.
`sharded_nullifier.ShardedNullifierSet` (imported on demand: it pulls in torch.distributed) spreads the nullifier
set over the GPUs of a node.
"""
from . import capi  # noqa: F401
from .capi import Engine, NullifierSet, ActError, load, build, LIB_PATH  # noqa: F401
from .api import (Params, PrivateKey, PublicKey, PreIssuance, IssuanceRequest, IssuanceResponse,  # noqa: F401
                  CreditToken, SpendProof, PreRefund, Refund, Error, CborError, NullifierDb, L)
