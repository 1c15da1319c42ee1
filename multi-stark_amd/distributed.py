"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

The reference has no distributed mode at all. Independent proofs shard one per rank with
no data-path collective; the only exchange is an all_gather of each rank's three 32-byte commitments
(stage 1, stage 2, quotient) that rank 0 folds into one joint digest — the "gather per-circuit commitments" step
of the north star. (A single joint Proof over one shared Merkle tree needs a row-range all-to-all before leaf
hashing, SURVEY §8e: not built in this round.)"""
import hashlib

import torch
import torch.distributed as dist


def rank_seeds(rank: int):
    """xorshift32 seeds of rank k's U32Add witness; rank 0 = the reference's bench witness."""
    return (0xDEADBEEF ^ ((rank * 0x9E3779B9) & 0xFFFFFFFF), 0xCAFEBABE ^ ((rank * 0x85EBCA6B) & 0xFFFFFFFF))


def commitments_of(proof: bytes, n_circuits: int) -> bytes:
    """stage_1 / stage_2 / quotient commitments out of Proof::to_bytes (active bitmap, then three caps)."""
    off = 8 + n_circuits
    out = []
    for _ in range(3):
        n = int.from_bytes(proof[off:off + 8], "little")
        out.append(proof[off + 8: off + 8 + 32 * n])
        off += 8 + 32 * n
    return b"".join(out)


def gather_commitments(commit: bytes, device=None):
    """all_gather of equal-length commitment blobs; returns the list ordered by rank (every rank gets it)."""
    world = dist.get_world_size()
    mine = torch.frombuffer(bytearray(commit), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [bytes(t.cpu().numpy().tobytes()) for t in out]


def joint_digest(commits) -> bytes:
    h = hashlib.blake2s()
    for i, c in enumerate(commits):
        h.update(i.to_bytes(4, "little"))
        h.update(c)
    return h.digest()
