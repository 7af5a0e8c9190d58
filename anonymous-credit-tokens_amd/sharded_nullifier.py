"""Nullifier set sharded over the GPUs of a node: the one step of the path with a real exchange.

The reference leaves the double-spend database to the caller (/root/reference/src/lib.rs:741-745; a
`HashSet<Scalar>` driven by `if is_spent(k) { reject } else { insert(k) }` in src/tests.rs:29-50).  On one GPU that
loop is `act_nullifier_check_and_insert_batch` (include/act_mi355x.h).  With one process per GPU, every rank verifies
its own shard of spend proofs, but a nullifier must be looked up where *all* earlier spends of it were recorded,
so the key space is partitioned: owner(k) = low 64 bits of (k mod l) mod world (nullifiers are uniform scalars; the
reduction makes k and k + l one key, as in the reference's HashSet<Scalar>).  A batch
call is then

    1. bucket this rank's nullifiers by owner (stable, on the GPU),
    2. all-to-all the bucket sizes, then the 32-byte keys           (RCCL over xGMI: 32 B per spend, once),
    3. check-and-insert what arrived in the local set, in (source rank, lane) order,
    4. all-to-all the one-byte answers back and undo the bucketing.

The result is what the sequential loop would give on the concatenation of all ranks' batches in rank order: among
equal nullifiers submitted in the same call, the lowest (rank, lane) is fresh and every other one is spent.

Host language note: the exchange is torch.distributed plumbing above the C ABI (backend "nccl" = RCCL on ROCm); the
set itself is the HIP open-addressing table of csrc/nullifier_impl.inc.  `local_set` may be injected (anything with
`check_and_insert_tensor(keys[m,32] uint8) -> spent[m] uint8`), which is how the two-rank gloo test drives the
exchange on CPU tensors without a GPU.
"""
from typing import Optional

import torch
import torch.distributed as dist


_ELL = 2**252 + 27742317777372353535851937790883648493


def reduce_mod_l(keys: torch.Tensor) -> torch.Tensor:
    """[n, 32] uint8 little-endian 256-bit values -> their canonical representatives mod l, same shape.  A nullifier is a
    SCALAR (the reference keeps a HashSet<Scalar>, and decode_scalar reduces, src/cbor.rs:85): k and k + l are one key, so
    the owner must be computed from -- and the owning shard must store -- the reduced value.  x < 2^256 < 16 l:
    with q = x >> 252 and l = 2^252 + c, x - q l = (x mod 2^252) - q c, plus l once if that went negative."""
    n = keys.shape[0]
    if n == 0:
        return keys.reshape(0, 32)
    w = keys.contiguous().view(torch.int32).reshape(n, 8).to(torch.int64) & 0xFFFFFFFF
    q = w[:, 7] >> 28
    limbs = [w[:, i].clone() for i in range(8)]
    limbs[7] = limbs[7] & 0x0FFFFFFF
    c = _ELL - 2**252
    qc, carry = [], torch.zeros_like(q)
    for i in range(4):
        t = q * ((c >> (32 * i)) & 0xFFFFFFFF) + carry
        qc.append(t & 0xFFFFFFFF); carry = t >> 32
    qc.append(carry)
    borrow = torch.zeros_like(q)
    for i in range(8):
        t = limbs[i] - (qc[i] if i < 5 else 0) - borrow
        borrow = (t < 0).to(torch.int64)
        limbs[i] = t + (borrow << 32)
    carry = torch.zeros_like(q)
    for i in range(8):                                  # + l where the difference was negative
        t = limbs[i] + borrow * ((_ELL >> (32 * i)) & 0xFFFFFFFF) + carry
        limbs[i] = t & 0xFFFFFFFF; carry = t >> 32
    cols = [((limbs[i] >> (8 * b)) & 0xFF) for i in range(8) for b in range(4)]
    return torch.stack(cols, dim=1).to(torch.uint8)


class _HipLocalSet:
    """Adapter: capi.NullifierSet on this rank's GPU, fed with device tensors."""

    def __init__(self, capacity: int, device: int, salt: Optional[bytes]):
        from . import capi
        self.set = capi.NullifierSet(capacity, device=device, salt=salt)
        self.device = device

    def check_and_insert_tensor(self, keys: torch.Tensor) -> torch.Tensor:
        m = keys.shape[0]
        out = torch.zeros(m, dtype=torch.uint8, device=keys.device)
        if m:
            keys = keys.contiguous()
            torch.cuda.synchronize(self.device)         # the set runs on its own stream: the received keys must have landed
            self.set.check_and_insert_dev(m, keys.data_ptr(), 32, 0, out.data_ptr())
        return out

    def __len__(self):
        return len(self.set)

    def close(self):
        self.set.close()


class ShardedNullifierSet:
    def __init__(self, capacity_per_rank: int, device: int = 0, salt: Optional[bytes] = None, group=None, local_set=None):
        if not dist.is_initialized():
            raise RuntimeError("ShardedNullifierSet needs an initialised torch.distributed process group (one rank per GPU)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.local = local_set if local_set is not None else _HipLocalSet(capacity_per_rank, device, salt)

    def owner(self, keys: torch.Tensor) -> torch.Tensor:
        """owner(k) = (k mod 2^63) mod world on the first 8 little-endian bytes: [n,32] uint8 -> [n] int64."""
        low = keys[:, :8].contiguous().view(torch.int64).reshape(-1)
        return (low & 0x7FFFFFFFFFFFFFFF) % self.world

    def check_and_insert(self, keys: torch.Tensor, skip_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """keys: [n,32] uint8 on this rank's device (n may differ per rank, 0 allowed); skip_mask: [n] uint8, non-zero =
        neither checked nor inserted (the status byte of a rejected proof).  Returns spent[n] uint8.  Collective: every
        rank of the group must call it."""
        n = keys.shape[0]
        dev = keys.device
        spent = torch.zeros(n, dtype=torch.uint8, device=dev)
        active = torch.arange(n, device=dev) if skip_mask is None else torch.nonzero(skip_mask == 0).reshape(-1)
        k = reduce_mod_l(keys[active]) if n else keys.reshape(0, 32)      # route and store scalars, not byte strings
        own = self.owner(k) if k.shape[0] else torch.zeros(0, dtype=torch.int64, device=dev)
        order = torch.argsort(own, stable=True)                       # lane order survives inside each owner's bucket
        send = k[order].contiguous()
        send_counts = torch.bincount(own, minlength=self.world).to(torch.int64)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        recv = torch.empty((sum(rc), 32), dtype=torch.uint8, device=dev)
        dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        ans = self.local.check_and_insert_tensor(recv)                # (source rank, lane) order = the sequential order
        back = torch.empty(send.shape[0], dtype=torch.uint8, device=dev)
        dist.all_to_all_single(back, ans.contiguous(), output_split_sizes=sc, input_split_sizes=rc, group=self.group)
        if back.shape[0]:
            spent[active[order]] = back
        return spent

    def __len__(self):
        """Nullifiers held by this rank's shard."""
        return len(self.local)

    def close(self):
        if hasattr(self.local, "close"):
            self.local.close()
