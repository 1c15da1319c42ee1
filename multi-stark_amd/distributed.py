"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

The reference has no distributed mode at all. Independent proofs shard one per rank with
no data-path collective; the only exchange is an all_gather of each rank's three 32-byte commitments
(stage 1, stage 2, quotient) that rank 0 folds into one joint digest — the "gather per-circuit commitments" step
of the north star. This is the "replicas" mode (bench.py's secondary figure). The single joint Proof over one shared Merkle
tree per commitment (row-range exchange before leaf hashing) is ms_prove_sharded, with its transports in
sharded.py / csrc/comm_rccl.hip / csrc/comm_local.hip."""
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


class CommitmentGatherer:
    """all_gather of per-proof commitment blobs off the proving thread. `submit` hands the blob to a worker thread
    and returns at once; the worker runs the collective on preallocated buffers, reads the result back and calls
    `on_gathered(list_of_blobs_by_rank)`. The prover (a ctypes call that releases the GIL) keeps the GPU busy with the
    next proof meanwhile, so the rank never stalls on the exchange. `finish` waits until everything submitted has
    been delivered and re-raises a worker error. One collective is in flight at a time, in submission order, so
    all ranks issue them in the same order."""

    def __init__(self, nbytes: int, device=None, on_gathered=None):
        import queue
        import threading

        self.world = dist.get_world_size()
        self.nbytes = nbytes
        self.device = device
        self.on_gathered = on_gathered
        self.results = []
        pin = device is not None
        self.host_in = torch.empty(nbytes, dtype=torch.uint8, pin_memory=pin)
        self.host_out = torch.empty(self.world * nbytes, dtype=torch.uint8, pin_memory=pin)
        if device is not None:
            self.dev_in = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self.dev_out = torch.empty(self.world * nbytes, dtype=torch.uint8, device=device)
        self.error = None
        self.q = queue.Queue()
        self.thread = threading.Thread(target=self._run, name="commitment-gather", daemon=True)
        self.thread.start()

    def _gather(self, commit: bytes):
        self.host_in.copy_(torch.frombuffer(bytearray(commit), dtype=torch.uint8))
        if self.device is not None:
            stream = self._stream
            with torch.cuda.stream(stream):
                self.dev_in.copy_(self.host_in, non_blocking=True)
                dist.all_gather_into_tensor(self.dev_out, self.dev_in)
                self.host_out.copy_(self.dev_out, non_blocking=True)
            stream.synchronize()
            flat = self.host_out
        else:
            outs = [torch.empty(self.nbytes, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(outs, self.host_in.clone())
            flat = torch.cat(outs)
        raw = flat.numpy().tobytes()
        return [raw[i * self.nbytes:(i + 1) * self.nbytes] for i in range(self.world)]

    def _run(self):
        if self.device is not None:
            torch.cuda.set_device(self.device)
            self._stream = torch.cuda.Stream(device=self.device)
        while True:
            item = self.q.get()
            try:
                if item is None:
                    return
                if self.error is None:
                    got = self._gather(item)
                    if self.on_gathered is not None:
                        self.on_gathered(got)
                    else:
                        self.results.append(got)
            except BaseException as e:  # surfaced by finish()
                self.error = e
            finally:
                self.q.task_done()

    def submit(self, commit: bytes):
        if len(commit) != self.nbytes:
            raise ValueError("commitment blob of %d bytes, expected %d" % (len(commit), self.nbytes))
        self.q.put(commit)

    def finish(self):
        self.q.join()
        if self.error is not None:
            raise self.error
        out, self.results = self.results, []
        return out

    def close(self):
        self.q.put(None)
        self.thread.join(timeout=60)


def joint_digest(commits) -> bytes:
    h = hashlib.blake2s()
    for i, c in enumerate(commits):
        h.update(i.to_bytes(4, "little"))
        h.update(c)
    return h.digest()
