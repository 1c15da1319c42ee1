"""Transports for `System.prove_sharded` (ms_prove_sharded): the two exchanges the library asks for, on torch.distributed
(`TorchComm`), on the library's own RCCL transport (`RcclComm`, csrc/comm_rccl.hip - no torch involved) or between threads
of one process (`LocalGroup` / `LocalComm`, csrc/comm_local.hip - device copies, no collective library at all).

`TorchComm` builds the `ms_comm` callback table of include/mstark.h. With backend "nccl" (= RCCL on ROCm) the device
buffers the library hands over are wrapped as torch tensors in place (`__cuda_array_interface__`) and exchanged by
`all_to_all_single` / `all_gather_into_tensor` over xGMI. With any other backend (gloo: tests and single-GPU
rehearsals, where several ranks share one device) the buffers are staged through host memory. Torch is plumbing here:
every byte that is exchanged was produced, and is consumed, by the library's own kernels."""
import ctypes as C

_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
_START = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t)
_WAIT = C.CFUNCTYPE(C.c_int32, C.c_void_p)
_ORDERED = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p)
_COLS = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t)
_SCATTER = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t)
_ABORT = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)
_COLS2 = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32)


class MsComm(C.Structure):
    _fields_ = [("size", C.c_uint32),  # sizeof(ms_comm) as this binding knows it: the library treats members beyond it as NULL
                ("rank", C.c_int32), ("world", C.c_int32), ("user", C.c_void_p), ("all_to_all", _CB), ("all_gather", _CB),
                ("all_to_all_start", _START), ("all_to_all_wait", _WAIT), ("all_to_all_cols_start", _COLS),
                ("set_stream_ordered", _ORDERED),  # (TorchComm leaves it NULL: its callbacks synchronise with the host)
                ("all_to_all_cols_start2", _COLS2),  # (NULL in TorchComm: the library then calls all_to_all_cols_start)
                ("scatter_cols_start", _SCATTER),
                ("abort", _ABORT)]  # (NULL in TorchComm: torch.distributed's own timeout is what ends a proof a rank has left)


class _DevBytes:
    """a device byte range as an object torch.as_tensor understands"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class TorchComm:
    def __init__(self, device_index=0, group=None, memory="device"):
        """memory="host": the pointers handed to the callbacks are HOST addresses (CPU tests of the exchange patterns,
        tests/test_distributed_cpu.py: the table's index arithmetic over gloo without a GPU); the library itself always
        passes device pointers."""
        global torch, dist
        import torch  # plumbing of this transport only: RcclComm below needs neither
        import torch.distributed as dist

        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.host_memory = memory == "host"
        self.device = torch.device("cpu") if self.host_memory else torch.device("cuda", device_index)
        self.direct = dist.get_backend(group) == "nccl" and not self.host_memory
        self.error = None
        self.bytes_moved = 0
        self._a2a = _CB(self._all_to_all)
        self._ag = _CB(self._all_gather)
        self._start = _START(self._all_to_all_start)
        self._wait = _WAIT(self._all_to_all_wait)
        self._cols = _COLS(self._all_to_all_cols_start)
        self._pending = []
        self._scatter = _SCATTER(self._scatter_cols_start)
        self.struct = MsComm(C.sizeof(MsComm), self.rank, self.world, None, self._a2a, self._ag, self._start, self._wait, self._cols)
        self.struct.scatter_cols_start = self._scatter

    def _view(self, ptr, nbytes):
        if self.host_memory:
            return torch.frombuffer((C.c_uint8 * nbytes).from_address(int(ptr)), dtype=torch.uint8)
        return torch.as_tensor(_DevBytes(ptr, nbytes), device=self.device)

    def _sync(self):
        if not self.host_memory:
            torch.cuda.synchronize(self.device)

    def _guard(self, fn):
        try:
            fn()
            return 0
        except BaseException as e:  # never unwind into C
            self.error = e
            return -1

    def reraise(self):
        if self.error is not None:
            e, self.error = self.error, None
            raise e

    def _all_to_all(self, _user, send, recv, per_peer):
        def run():
            n = per_peer * self.world
            s, r = self._view(send, n), self._view(recv, n)
            self.bytes_moved += n
            if self.direct:
                dist.all_to_all_single(r, s, group=self.group)
                self._sync()
            else:
                hs = s.cpu()
                parts = [torch.empty(n, dtype=torch.uint8) for _ in range(self.world)]
                dist.all_gather(parts, hs, group=self.group)  # gloo has no all_to_all on every build: take my block of each
                hr = torch.cat([p[self.rank * per_peer:(self.rank + 1) * per_peer] for p in parts])
                r.copy_(hr)
                self._sync()

        return self._guard(run)

    def _all_to_all_start(self, _user, send, send_stride, recv, recv_stride, per_peer):
        """non-blocking exchange of one column group: chunk k lives at send + k * send_stride / recv + k * recv_stride"""

        def run():
            ins = [self._view(send + k * send_stride, per_peer) for k in range(self.world)]
            outs = [self._view(recv + k * recv_stride, per_peer) for k in range(self.world)]
            self.bytes_moved += per_peer * self.world
            if self.direct:
                self._pending.append(dist.all_to_all(outs, ins, group=self.group, async_op=True))  # RCCL's own stream
            else:  # staged transports have nothing to overlap with: exchange now
                mine = torch.cat([t.cpu() for t in ins])
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=self.group)
                for k in range(self.world):
                    outs[k].copy_(parts[k][self.rank * per_peer:(self.rank + 1) * per_peer])
                self._sync()

        return self._guard(run)

    def _all_to_all_cols_start(self, _user, send, sps, scs, recv, rps, rcs, ncols, seg):
        """the exchange read straight out of a column-major matrix: for rank k, segment c is sent from send + k * sps + c * scs
        and the segment c coming from rank k is written at recv + k * rps + c * rcs"""

        def run():
            self.bytes_moved += seg * ncols * self.world
            ins = [[self._view(send + k * sps + c * scs, seg) for c in range(ncols)] for k in range(self.world)]
            outs = [[self._view(recv + k * rps + c * rcs, seg) for c in range(ncols)] for k in range(self.world)]
            for c in range(ncols):  # this rank's own rows never leave the device
                outs[self.rank][c].copy_(ins[self.rank][c])
            if self.direct:
                ops = []
                for k in range(self.world):
                    if k == self.rank:
                        continue
                    for c in range(ncols):
                        ops.append(dist.P2POp(dist.isend, ins[k][c], k, self.group))
                        ops.append(dist.P2POp(dist.irecv, outs[k][c], k, self.group))
                if ops:
                    self._pending.extend(dist.batch_isend_irecv(ops))
            else:  # staged transports have nothing to overlap with: exchange now (every rank takes its block of each)
                mine = torch.cat([t.cpu() for k in range(self.world) for t in ins[k]])
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=self.group)
                per = seg * ncols
                for k in range(self.world):
                    if k == self.rank:
                        continue
                    blk = parts[k][self.rank * per:(self.rank + 1) * per]
                    for c in range(ncols):
                        outs[k][c].copy_(blk[c * seg:(c + 1) * seg])
                self._sync()

        return self._guard(run)

    def _scatter_cols_start(self, _user, root, send, sps, scs, recv, rcs, ncols, seg):
        """one rank's matrix handed out by row ranges: root sends ncols segments to every other rank (general ownership)"""

        def run():
            if self.rank == root:
                self.bytes_moved += seg * ncols * (self.world - 1)
                ins = [[self._view(send + k * sps + c * scs, seg) for c in range(ncols)] for k in range(self.world)]
            else:
                self.bytes_moved += seg * ncols
                outs = [self._view(recv + c * rcs, seg) for c in range(ncols)]
            if self.direct:
                ops = []
                if self.rank == root:
                    for k in range(self.world):
                        if k != root:
                            ops += [dist.P2POp(dist.isend, ins[k][c], k, self.group) for c in range(ncols)]
                else:
                    ops = [dist.P2POp(dist.irecv, outs[c], root, self.group) for c in range(ncols)]
                if ops:
                    self._pending.extend(dist.batch_isend_irecv(ops))
            else:  # staged: the root's whole matrix is broadcast through the host, every rank takes its range
                per = seg * ncols
                buf = torch.cat([t.cpu() for k in range(self.world) for t in ins[k]]) if self.rank == root else torch.empty(per * self.world, dtype=torch.uint8)
                dist.broadcast(buf, src=root, group=self.group)
                if self.rank != root:
                    blk = buf[self.rank * per:(self.rank + 1) * per]
                    for c in range(ncols):
                        outs[c].copy_(blk[c * seg:(c + 1) * seg])
                    self._sync()

        return self._guard(run)

    def _all_to_all_wait(self, _user):
        def run():
            for w in self._pending:
                w.wait()
            self._pending = []
            self._sync()

        return self._guard(run)

    def _all_gather(self, _user, send, recv, nbytes):
        def run():
            s, r = self._view(send, nbytes), self._view(recv, nbytes * self.world)
            self.bytes_moved += nbytes * self.world
            if self.direct:
                dist.all_gather_into_tensor(r, s, group=self.group)
                self._sync()
            else:
                parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
                dist.all_gather(parts, s.cpu(), group=self.group)
                r.copy_(torch.cat(parts))
                self._sync()

        return self._guard(run)


class RcclComm:
    """The library's own transport (ms_comm_rccl_*, csrc/comm_rccl.hip): grouped ncclSend / ncclRecv and ncclAllGather called
    from C, nothing of the exchange passes through Python. `unique_id`: the 128 bytes of `RcclComm.unique_id()` drawn on
    rank 0 and handed to every rank by the host's own channel (bench.py broadcasts them with torch.distributed)."""

    def __init__(self, ctx, unique_id: bytes, rank: int, world: int):
        from . import lib, _check

        self._lib = lib()
        self.ctx = ctx
        self.rank, self.world = rank, world
        self.h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id if unique_id else bytes(128))
        _check(self._lib.ms_comm_rccl_create(ctx.h, buf, C.c_int32(rank), C.c_int32(world), C.byref(self.h)))
        self._lib.ms_comm_rccl_table.restype = C.c_void_p
        self._lib.ms_comm_rccl_bytes_moved.restype = C.c_uint64
        self.struct = MsComm.from_address(self._lib.ms_comm_rccl_table(self.h))
        self._base = 0

    @staticmethod
    def unique_id() -> bytes:
        from . import lib, _check

        buf = (C.c_uint8 * 128)()
        _check(lib().ms_comm_rccl_unique_id(buf))
        return bytes(buf)

    @property
    def bytes_moved(self):
        return int(self._lib.ms_comm_rccl_bytes_moved(self.h)) - self._base

    @bytes_moved.setter
    def bytes_moved(self, v):
        self._base = int(self._lib.ms_comm_rccl_bytes_moved(self.h)) - int(v)

    def reraise(self):
        pass  # errors of this transport come back as the library's error codes

    def close(self):
        if getattr(self, "h", None):
            self._lib.ms_comm_rccl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LocalGroup:
    """The rendezvous of the library's in-process transport (ms_comm_local_*, csrc/comm_local.hip): the ranks are threads of
    this process, each with its own Context; the exchanges are device copies ordered by HIP events. `run(fn)` starts one
    thread per rank, calls fn(rank, group) on each and returns the results in rank order; a rank that raises aborts the
    group, so that the others return with an error instead of waiting for it."""

    def __init__(self, world: int):
        from . import lib, _check

        self._lib = lib()
        self.world = world
        self.h = C.c_void_p()
        _check(self._lib.ms_comm_local_group_create(C.c_int32(world), C.byref(self.h)))

    def comm(self, ctx, rank: int):
        return LocalComm(self, ctx, rank)

    def abort(self):
        if getattr(self, "h", None):
            self._lib.ms_comm_local_group_abort(self.h)

    def run(self, fn):
        import threading

        results, errors = [None] * self.world, [None] * self.world

        def body(r):
            try:
                results[r] = fn(r, self)
            except BaseException as e:  # noqa: BLE001 - reported by the caller's thread
                errors[r] = e
                self.abort()

        threads = [threading.Thread(target=body, args=(r,), name="ms-rank-%d" % r) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        first = [e for e in errors if e is not None and "local transport aborted" not in str(e)] or [e for e in errors if e is not None]
        if first:
            raise first[0]
        return results

    def close(self):
        if getattr(self, "h", None):
            self._lib.ms_comm_local_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LocalComm:
    """One rank's handle of the in-process transport; `struct` is the ms_comm table ms_prove_sharded takes."""

    def __init__(self, group: LocalGroup, ctx, rank: int):
        from . import lib, _check

        self._lib = lib()
        self.group, self.ctx = group, ctx
        self.rank, self.world = rank, group.world
        self.h = C.c_void_p()
        _check(self._lib.ms_comm_local_create(group.h, ctx.h, C.c_int32(rank), C.byref(self.h)))
        self._lib.ms_comm_local_table.restype = C.c_void_p
        self._lib.ms_comm_local_bytes_moved.restype = C.c_uint64
        self.struct = MsComm.from_address(self._lib.ms_comm_local_table(self.h))

    @property
    def bytes_moved(self):
        return int(self._lib.ms_comm_local_bytes_moved(self.h))

    def reraise(self):
        pass  # errors of this transport come back as the library's error codes

    def close(self):
        if getattr(self, "h", None):
            self._lib.ms_comm_local_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def u32_add_owners(k):
    """owners for the system [ByteTable, U32Add x k]: the byte table replicated, adder i on rank i."""
    return [-1] + list(range(k))
