"""multi-stark hot path on MI355X: Python host mirror of the reference's interface over the C ABI.

Names follow the reference: `System.new(config, circuits)` (src/system.rs:115), `SystemWitness.from_stage_1`
(src/system.rs:244), `System.prove_multiple_claims` (src/prover.rs:290), `Proof.to_bytes` (src/prover.rs:245).
All computation happens in `libmstark_hip.so` (hand-written gfx950 kernels behind include/mstark.h); there is no
CPU fallback: loading fails loudly if the library or a HIP device is missing.
"""
import ctypes as C
import os

import numpy as np

from . import frontend  # noqa: F401
from . import babybear  # noqa: F401  (the reference's second configuration, include/mstark_bb.h)
from .frontend import Params, bench_params, test_params, compile_circuit, system_blob, pack_claims  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmstark_hip.so")

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
_lib = None


class MstarkError(RuntimeError):
    pass


def lib():
    """Load the HIP library (never a substitute): raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MstarkError("libmstark_hip.so is missing: run `python multi-stark_amd/build.py` (needs hipcc)")
        L = C.CDLL(LIB_PATH)
        L.ms_last_error.restype = C.c_char_p
        L.ms_kernel_name.restype = C.c_char_p
        _lib = L
    return _lib


def exported_symbols():
    """Every entry point include/mstark.h declares (used by the CPU-side ABI test)."""
    return ["ms_last_error", "ms_device_count", "ms_ctx_create", "ms_ctx_destroy", "ms_ctx_sync", "ms_ctx_sync_count", "ms_ctx_trim", "ms_ctx_set_profile_mask",
            "ms_ctx_kernel_stats", "ms_ctx_kernel_units", "ms_ctx_reset_stats", "ms_ctx_debug_fail_alloc", "ms_kernel_count", "ms_kernel_name", "ms_system_create",
            "ms_system_destroy", "ms_system_preprocessed_commit", "ms_system_circuit_info", "ms_witness_create", "ms_witness_create_host", "ms_claims_slice_range", "ms_witness_create_host_sliced", "ms_witness_prefetch",
            "ms_witness_u32_add_bench", "ms_witness_destroy", "ms_prove", "ms_prove_sharded", "ms_ctx_comm_progress", "ms_comm_rccl_unique_id", "ms_comm_rccl_create",
            "ms_comm_rccl_table", "ms_comm_rccl_bytes_moved", "ms_comm_rccl_destroy", "ms_comm_local_group_create", "ms_comm_local_group_abort",
            "ms_comm_local_group_destroy", "ms_comm_local_create", "ms_comm_local_table", "ms_comm_local_bytes_moved", "ms_comm_local_destroy", "ms_verify", "ms_dft_batch", "ms_coset_lde_batch", "ms_quotient_lde", "ms_mmcs_commit",
            "ms_mmcs_open", "ms_mmcs_destroy", "ms_blake3", "ms_pcs_commit", "ms_pcs_open", "ms_pcs_verify", "ms_challenger_create",
            "ms_challenger_destroy", "ms_challenger_observe", "ms_challenger_observe_digests", "ms_challenger_sample_ext",
            "ms_challenger_sample_bits", "ms_stage2_trace", "ms_claims_accumulator",
            "ms_quotient_values", "ms_field_op", "ms_trace_destroy", "ms_trace_info", "ms_system_preprocessed_mmcs",
            "ms_witness_commit_stage1", "ms_challenger_observe_claims", "ms_witness_claims_accumulator", "ms_stage2_build",
            "ms_pcs_commit_traces", "ms_quotient", "ms_pcs_commit_ldes"]


def _check(rc):
    if rc != 0:
        raise MstarkError(lib().ms_last_error().decode() or ("mstark error %d" % rc))


def _p(a):
    return a.ctypes.data_as(u64p)


def _b(a):
    return a.ctypes.data_as(u8p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def device_count() -> int:
    """HIP devices visible to this process (ms_device_count; 0 without a GPU or driver)"""
    return int(lib().ms_device_count())


class Context:
    """One HIP device (ms_ctx)."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        rc = lib().ms_ctx_create(C.c_int32(device), C.byref(self.h))
        if rc != 0:
            raise MstarkError("cannot create a HIP context: " + lib().ms_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            lib().ms_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(lib().ms_ctx_sync(self.h))

    def sync_count(self):
        """host waits on this context's stream so far (ms_ctx_sync_count): a proof's count is the difference around it"""
        n = C.c_uint64()
        _check(lib().ms_ctx_sync_count(self.h, C.byref(n)))
        return int(n.value)

    def trim(self):
        _check(lib().ms_ctx_trim(self.h))

    def comm_progress(self):
        """(text of the last transport call ms_prove_sharded entered through this context, calls entered so far, still inside?) -
        readable from another thread while the proof runs"""
        buf = C.create_string_buffer(256)
        seq, fl = C.c_uint64(), C.c_int32()
        _check(lib().ms_ctx_comm_progress(self.h, buf, C.c_size_t(256), C.byref(seq), C.byref(fl)))
        return buf.value.decode(), int(seq.value), bool(fl.value)

    def debug_fail_alloc(self, nth):
        """diagnostics: the nth device allocation from now raises (0 = off)"""
        _check(lib().ms_ctx_debug_fail_alloc(self.h, C.c_int32(nth)))

    # ---- profiling
    def kernel_names(self):
        return [lib().ms_kernel_name(C.c_int32(i)).decode() for i in range(lib().ms_kernel_count())]

    def set_profile(self, names):
        all_names = self.kernel_names()
        mask = 0
        for n in names:
            mask |= 1 << all_names.index(n)
        _check(lib().ms_ctx_set_profile_mask(self.h, C.c_uint32(mask)))

    def reset_stats(self):
        _check(lib().ms_ctx_reset_stats(self.h))

    def kernel_stats(self):
        out = {}
        for i, n in enumerate(self.kernel_names()):
            launches, ms, byts = C.c_uint64(), C.c_double(), C.c_double()
            _check(lib().ms_ctx_kernel_stats(self.h, C.c_int32(i), C.byref(launches), C.byref(ms), C.byref(byts)))
            units = C.c_double()
            _check(lib().ms_ctx_kernel_units(self.h, C.c_int32(i), C.byref(units)))
            out[n] = {"launches": launches.value, "ms": ms.value, "alg_bytes": byts.value, "units": units.value}
        return out

    # ---- PCS-level entry points
    def dft_batch(self, m, inverse=False):
        m = _u64(m)
        out = np.empty_like(m)
        _check(lib().ms_dft_batch(self.h, _p(m), C.c_size_t(m.shape[0]), C.c_size_t(m.shape[1]), C.c_int32(int(inverse)), _p(out)))
        return out

    def coset_lde_batch(self, m, log_blowup):
        m = _u64(m)
        out = np.empty((m.shape[0] << log_blowup, m.shape[1]), dtype=np.uint64)
        _check(lib().ms_coset_lde_batch(self.h, _p(m), C.c_size_t(m.shape[0]), C.c_size_t(m.shape[1]), C.c_uint32(log_blowup), _p(out)))
        return out

    def quotient_lde(self, q, log_n, log_q, log_blowup):
        q = _u64(q)
        D = q.shape[1]
        out = np.empty((1 << (log_n + log_blowup), D << log_q), dtype=np.uint64)
        _check(lib().ms_quotient_lde(self.h, _p(q), C.c_uint32(log_n), C.c_uint32(log_q), C.c_uint32(log_blowup), C.c_size_t(D), _p(out)))
        return out

    def blake3(self, data: bytes) -> bytes:
        buf = np.frombuffer(data, dtype=np.uint8) if data else np.zeros(1, dtype=np.uint8)
        out = np.zeros(32, dtype=np.uint8)
        _check(lib().ms_blake3(self.h, _b(buf), C.c_size_t(len(data)), _b(out)))
        return out.tobytes()

    def stage2_trace(self, mult, arg_offsets, args, beta, gamma, acc_in):
        mult, args, arg_offsets = _u64(mult), _u64(args), _u64(arg_offsets)
        h, L = mult.shape
        tr = np.zeros((h, max(L, 1) * 2), dtype=np.uint64)
        acc = np.zeros(2, dtype=np.uint64)
        _check(lib().ms_stage2_trace(self.h, C.c_size_t(h), C.c_size_t(L), _p(mult), _p(arg_offsets), _p(args), _p(_u64(beta)),
                                     _p(_u64(gamma)), _p(_u64(acc_in)), _p(tr), _p(acc)))
        return tr, (int(acc[0]), int(acc[1]))

    def claims_accumulator(self, claims_packed, beta, gamma):
        offs, data = claims_packed
        data = data if data.size else np.zeros(1, dtype=np.uint64)
        acc = np.zeros(2, dtype=np.uint64)
        _check(lib().ms_claims_accumulator(self.h, C.c_size_t(len(offs) - 1), _p(offs), _p(data), _p(_u64(beta)), _p(_u64(gamma)), _p(acc)))
        return int(acc[0]), int(acc[1])

    def field_op(self, op, a, b=None):
        a = _u64(a)
        bb = _u64(b) if b is not None else None
        out = np.empty_like(a)
        n = a.size // 2 if op >= 4 else a.size
        _check(lib().ms_field_op(self.h, C.c_int32(op), _p(a), _p(bb) if bb is not None else None, C.c_size_t(n), _p(out)))
        return out


class Mmcs:
    """ProverData of one MerkleTreeMmcs commitment (ms_mmcs)."""

    def __init__(self, ctx, mats, cap_height=0):
        self.ctx = ctx
        self.mats = [_u64(m) for m in mats]
        n = len(self.mats)
        ptrs = (u64p * n)(*[_p(m) for m in self.mats])
        hs = _u64([m.shape[0] for m in self.mats])
        ws = _u64([m.shape[1] for m in self.mats])
        maxh = int(hs.max())
        cap = np.zeros(32 * min(1 << cap_height, maxh), dtype=np.uint8)
        self.h = C.c_void_p()
        _check(lib().ms_mmcs_commit(ctx.h, C.c_size_t(n), ptrs, _p(hs), _p(ws), C.c_uint32(cap_height), _b(cap), C.byref(self.h)))
        self.cap = cap.tobytes()
        self.log_max = maxh.bit_length() - 1
        self.widths = ws

    def open(self, index):
        vals = np.zeros(int(self.widths.sum()), dtype=np.uint64)
        proof = np.zeros(32 * (self.log_max + 1), dtype=np.uint8)
        ns = C.c_size_t()
        _check(lib().ms_mmcs_open(self.h, C.c_size_t(index), _p(vals), _b(proof), C.byref(ns)))
        return vals, proof[: 32 * ns.value].tobytes()

    def __del__(self):
        if getattr(self, "h", None):
            lib().ms_mmcs_destroy(self.h)
            self.h = None


class DeviceCommitment(Mmcs):
    """ProverData produced by a Level-2 call (ms_witness_commit_stage1, ms_pcs_commit_traces, ms_pcs_commit_ldes,
    ms_system_preprocessed_mmcs): the matrices never left the device."""

    def __init__(self, ctx, handle, cap: bytes, widths, log_max):
        self.ctx, self.h, self.cap = ctx, handle, cap
        self.widths = _u64(widths)
        self.log_max = log_max
        self.mats = []


class Trace:
    """A device-resident matrix handed between Level-2 calls (ms_trace): stage-2 evaluations or a quotient LDE."""

    def __init__(self, handle):
        self.h = handle

    def info(self):
        o = np.zeros(3, dtype=np.uint64)
        _check(lib().ms_trace_info(self.h, _p(o)))
        return int(o[0]), int(o[1]), int(o[2])

    def __del__(self):
        if getattr(self, "h", None):
            lib().ms_trace_destroy(self.h)
            self.h = None


def _cap_buffer(cap_height, max_height):
    return np.zeros(32 * min(1 << cap_height, max_height), dtype=np.uint8)


def pcs_commit_traces(ctx, traces, log_blowup, cap_height=0):
    """Pcs::commit (src/prover.rs:414-419) on evaluation handles; the handles are consumed"""
    infos = [t.info() for t in traces]
    maxh = max(h for h, _, _ in infos) << log_blowup
    cap = _cap_buffer(cap_height, maxh)
    hs = (C.c_void_p * len(traces))(*[t.h for t in traces])
    out = C.c_void_p()
    _check(lib().ms_pcs_commit_traces(ctx.h, C.c_uint32(log_blowup), C.c_uint32(cap_height), C.c_size_t(len(traces)), hs, _b(cap), C.byref(out)))
    return DeviceCommitment(ctx, out, cap.tobytes(), [w for _, w, _ in infos], maxh.bit_length() - 1)


def pcs_commit_ldes(ctx, ldes, cap_height=0):
    """Pcs::commit_ldes (src/prover.rs:526) on LDE handles; the commitment takes the matrices over"""
    infos = [t.info() for t in ldes]
    maxh = max(h for h, _, _ in infos)
    cap = _cap_buffer(cap_height, maxh)
    hs = (C.c_void_p * len(ldes))(*[t.h for t in ldes])
    out = C.c_void_p()
    _check(lib().ms_pcs_commit_ldes(ctx.h, C.c_uint32(cap_height), C.c_size_t(len(ldes)), hs, _b(cap), C.byref(out)))
    return DeviceCommitment(ctx, out, cap.tobytes(), [w for _, w, _ in infos], maxh.bit_length() - 1)


class PcsCommitment(Mmcs):
    """Pcs::commit (examples/pcs_example.rs:64-69): evaluations over the natural domains -> coset LDE + Merkle tree on the device."""

    def __init__(self, ctx, evals, log_blowup, cap_height=0):
        self.ctx = ctx
        self.mats = [_u64(m) for m in evals]
        n = len(self.mats)
        ptrs = (u64p * n)(*[_p(m) for m in self.mats])
        hs = _u64([m.shape[0] for m in self.mats])
        ws = _u64([m.shape[1] for m in self.mats])
        maxh = int(hs.max()) << log_blowup
        cap = np.zeros(32 * min(1 << cap_height, maxh), dtype=np.uint8)
        self.h = C.c_void_p()
        _check(lib().ms_pcs_commit(ctx.h, C.c_uint32(log_blowup), C.c_uint32(cap_height), C.c_size_t(n), ptrs, _p(hs), _p(ws), _b(cap),
                                   C.byref(self.h)))
        self.cap = cap.tobytes()
        self.log_max = maxh.bit_length() - 1
        self.widths = ws
        self.log_n = [int(h).bit_length() - 1 for h in hs]


class Challenger:
    """config.initialise_challenger() as a handle (ms_challenger): src/types.rs:28-81,118-130"""

    def __init__(self, params):
        self.h = C.c_void_p()
        _check(lib().ms_challenger_create(_p(_u64(params.words())), C.byref(self.h)))

    def observe(self, elems):
        e = np.atleast_1d(np.asarray(elems, dtype=np.uint64))  # dtype first: a tuple of words above 2^63 would pass through float64
        _check(lib().ms_challenger_observe(self.h, _p(e), C.c_size_t(e.size)))

    def observe_digests(self, cap: bytes):
        a = np.frombuffer(cap, dtype=np.uint8)
        _check(lib().ms_challenger_observe_digests(self.h, _b(a), C.c_size_t(len(cap) // 32)))

    def observe_claims(self, witness):
        """the claims of a device-resident witness, absorbed as src/prover.rs:369-373 does (hashed on the device when long)"""
        _check(lib().ms_challenger_observe_claims(self.h, witness.h))

    def sample_ext(self):
        o = np.zeros(2, dtype=np.uint64)
        _check(lib().ms_challenger_sample_ext(self.h, _p(o)))
        return int(o[0]), int(o[1])

    def sample_bits(self, bits):
        o = C.c_uint64()
        _check(lib().ms_challenger_sample_bits(self.h, C.c_uint32(bits), C.byref(o)))
        return o.value

    def __del__(self):
        if getattr(self, "h", None):
            lib().ms_challenger_destroy(self.h)
            self.h = None


def _flatten_points(per_round):
    n_points, pts = [], []
    for per_matrix in per_round:
        for plist in per_matrix:
            n_points.append(len(plist))
            for z in plist:
                pts.extend(int(x) for x in z)
    return _u64(n_points), _u64(pts if pts else [0])


def pcs_open(ctx, params, rounds, challenger):
    """Pcs::open (src/prover.rs:580). rounds: [(PcsCommitment | Mmcs, [[(c0, c1), ..] per matrix])].
    Returns (opened values flat, round -> matrix -> point -> column, 2 words each; FriProof bytes)."""
    n_points, pts = _flatten_points([per for _, per in rounds])
    handles = (C.c_void_p * len(rounds))(*[m.h for m, _ in rounds])
    total = sum(len(plist) * int(m.widths[i]) for m, per in rounds for i, plist in enumerate(per))
    opened = np.zeros(max(total, 1) * 2, dtype=np.uint64)
    cap = 1 << 22
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_size_t()
    _check(lib().ms_pcs_open(ctx.h, _p(_u64(params.words())), C.c_size_t(len(rounds)), handles, _p(n_points), _p(pts), challenger.h, _p(opened),
                             C.c_size_t(opened.size), _b(out), C.c_size_t(cap), C.byref(n)))
    return opened[: total * 2].copy(), out[: n.value].tobytes()


def pcs_verify(params, rounds, opened, fri: bytes, challenger):
    """Pcs::verify. rounds: [(cap bytes, [(log_n, width)] per matrix, [[point, ..] per matrix])]; True = accepted"""
    caps = [np.frombuffer(c, dtype=np.uint8) for c, _, _ in rounds]
    cap_ptrs = (u8p * len(rounds))(*[_b(c) for c in caps])
    cap_sizes = _u64([len(c) // 32 for c, _, _ in rounds])
    n_mats = _u64([len(d) for _, d, _ in rounds])
    log_n = _u64([ln for _, d, _ in rounds for ln, _w in d])
    widths = _u64([w for _, d, _ in rounds for _ln, w in d])
    n_points, pts = _flatten_points([per for _, _, per in rounds])
    f = np.frombuffer(fri, dtype=np.uint8) if fri else np.zeros(1, dtype=np.uint8)
    op = _u64(opened) if len(opened) else np.zeros(1, dtype=np.uint64)
    ok = C.c_int32()
    _check(lib().ms_pcs_verify(_p(_u64(params.words())), C.c_size_t(len(rounds)), cap_ptrs, _p(cap_sizes), _p(n_mats), _p(log_n), _p(widths),
                               _p(n_points), _p(pts), _p(op), _b(f), C.c_size_t(len(fri)), challenger.h, C.byref(ok)))
    return ok.value == 1


class Proof:
    """Serialized `Proof<GoldilocksBlake3Config>` (src/prover.rs:213-254)."""

    def __init__(self, data: bytes, stage_ms=None):
        self._bytes = data
        self.stage_ms = stage_ms

    def to_bytes(self) -> bytes:
        return self._bytes

    @staticmethod
    def from_bytes(data: bytes):
        return Proof(bytes(data))


class SystemWitness:
    """Device-resident SystemWitness + claims (ms_witness). Built by `System.witness`."""

    def __init__(self, handle, rows, system):
        self.h = handle
        self.rows = rows  # sum of active trace heights
        self.system = system  # keeps System (and its Context) alive: device buffers return to that context's pool

    def __del__(self):
        if getattr(self, "h", None):
            lib().ms_witness_destroy(self.h)
            self.h = None

    def prefetch(self, on=True):
        """host-resident witness: every proof also uploads the inputs of the next one while it computes (ms_witness_prefetch)"""
        _check(lib().ms_witness_prefetch(self.h, C.c_int32(1 if on else 0)))

    # ---- Level 2 (include/mstark.h): the prover's steps one by one, everything staying on the device
    def commit_stage1(self, heights, widths):
        """pcs.commit of the stage-1 traces (src/prover.rs:338-350); heights / widths of the ACTIVE circuits"""
        sysm = self.system
        lb, ch = sysm.params.log_blowup, sysm.params.cap_height
        maxh = max(heights) << lb
        cap = _cap_buffer(ch, maxh)
        out = C.c_void_p()
        _check(lib().ms_witness_commit_stage1(self.h, _b(cap), C.byref(out)))
        return DeviceCommitment(sysm.ctx, out, cap.tobytes(), widths, maxh.bit_length() - 1)

    def claims_accumulator(self, beta, gamma):
        acc = np.zeros(2, dtype=np.uint64)
        _check(lib().ms_witness_claims_accumulator(self.h, _p(_u64(beta)), _p(_u64(gamma)), _p(acc)))
        return int(acc[0]), int(acc[1])

    def stage2_build(self, n_active, beta, gamma, acc_in):
        """LookupValues::stage_2_traces (src/prover.rs:400): ([accumulator after each active circuit], [Trace handles])"""
        accs = np.zeros(2 * n_active, dtype=np.uint64)
        hs = (C.c_void_p * n_active)()
        _check(lib().ms_stage2_build(self.h, _p(_u64(beta)), _p(_u64(gamma)), _p(_u64(acc_in)), _p(accs), hs))
        return [(int(accs[2 * i]), int(accs[2 * i + 1])) for i in range(n_active)], [Trace(C.c_void_p(h)) for h in hs]


class System:
    """System<GoldilocksBlake3Config> + ProverKey on one device."""

    def __init__(self, ctx, blob: bytes, n_circuits: int):
        self.ctx = ctx
        self.n_circuits = n_circuits
        self.blob = blob
        a = np.frombuffer(blob, dtype=np.uint8)
        self.h = C.c_void_p()
        _check(lib().ms_system_create(ctx.h, _b(a), C.c_size_t(len(blob)), C.byref(self.h)))

    @staticmethod
    def new(ctx, params, circuit_inputs):
        """`System::new(config, inputs)`: compiles each circuit with the front-end and commits the preprocessed traces."""
        compiled = [compile_circuit(ci) for ci in circuit_inputs]
        s = System(ctx, system_blob(params, compiled), len(compiled))
        s.params = params
        return s

    def __del__(self):
        if getattr(self, "h", None):
            lib().ms_system_destroy(self.h)
            self.h = None

    def preprocessed_commit(self):
        out = np.zeros(32 * 256, dtype=np.uint8)
        n = C.c_size_t()
        _check(lib().ms_system_preprocessed_commit(self.h, _b(out), C.c_size_t(out.size), C.byref(n)))
        return out[: 32 * n.value].tobytes() if n.value else None

    def circuit_info(self, ci):
        o = np.zeros(9, dtype=np.uint64)
        _check(lib().ms_system_circuit_info(self.h, C.c_size_t(ci), _p(o)))
        keys = ["main_width", "pre_width", "pre_height", "num_lookups", "stage2_width", "constraint_count",
                "max_constraint_degree", "quotient_degree", "args_width"]
        return dict(zip(keys, (int(x) for x in o)))

    def witness(self, traces, claims_packed, lookups=None, remote_heights=None):
        """`SystemWitness::from_stage_1` (lookups=None) or an explicit SystemWitness{traces, lookups}; uploads to HBM.
        remote_heights {circuit: height}: circuits another rank computes (prove_sharded): no trace here, height only."""
        remote_heights = remote_heights or {}
        trs = [_u64(t) if t is not None and len(t) else np.zeros((0, 1), dtype=np.uint64) for t in traces]
        n = self.n_circuits
        if len(trs) != n:
            raise MstarkError("expected one trace per circuit")
        tptr = (u64p * n)(*[None if i in remote_heights else _p(t) for i, t in enumerate(trs)])
        hs = _u64([remote_heights.get(i, t.shape[0]) for i, t in enumerate(trs)])
        mptr = aptr = None
        keep = []
        if lookups is not None:
            ms, as_ = [], []
            for (m, a) in lookups:
                m, a = _u64(m), _u64(a)
                keep += [m, a]
                ms.append(_p(m))
                as_.append(_p(a))
            mptr, aptr = (u64p * n)(*ms), (u64p * n)(*as_)
        offs, data = claims_packed
        data = data if data.size else np.zeros(1, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().ms_witness_create(self.h, tptr, _p(hs), mptr, aptr, C.c_size_t(len(offs) - 1), _p(offs), _p(data), C.byref(h)))
        return SystemWitness(h, int(hs.sum()), self)

    def host_witness(self, traces, claims_packed, remote_heights=None):
        """A SystemWitness that stays in HOST memory (ms_witness_create_host): every prove_multiple_claims uploads it,
        as the reference's prove() would receive it (benches/multi_stark.rs:292-296). The arrays are kept alive (and
        page-locked) by the returned object. remote_heights {circuit: height}: circuits another rank computes
        (prove_sharded): no trace here, height only."""
        remote_heights = remote_heights or {}
        trs = [_u64(t) if t is not None and len(t) else np.zeros((0, 1), dtype=np.uint64) for t in traces]
        n = self.n_circuits
        if len(trs) != n:
            raise MstarkError("expected one trace per circuit")
        tptr = (u64p * n)(*[None if i in remote_heights else _p(t) for i, t in enumerate(trs)])
        hs = _u64([remote_heights.get(i, t.shape[0]) for i, t in enumerate(trs)])
        offs, data = claims_packed
        data = data if data.size else np.zeros(1, dtype=np.uint64)
        h = C.c_void_p()
        pinned = C.c_int32(0)
        _check(lib().ms_witness_create_host(self.h, tptr, _p(hs), C.c_size_t(len(offs) - 1), _p(offs), _p(data), C.byref(pinned), C.byref(h)))
        w = SystemWitness(h, int(hs.sum()), self)
        w.keep = trs
        w.pinned = bool(pinned.value)
        return w

    def claims_slice_range(self, heights, claim_offsets, rank, world):
        """(first element, count) of the claims' data that rank `rank` of `world` reads in a joint proof (ms_claims_slice_range)"""
        hs, offs = _u64(heights), _u64(claim_offsets)
        first, count = C.c_uint64(), C.c_uint64()
        _check(lib().ms_claims_slice_range(self.h, _p(hs), C.c_size_t(len(offs) - 1), _p(offs), C.c_int32(rank), C.c_int32(world), C.byref(first), C.byref(count)))
        return int(first.value), int(count.value)

    def host_witness_sliced(self, traces, claim_offsets, data_first, data_slice, head, remote_heights=None):
        """host_witness for a rank of a joint proof that holds only ITS part of the claims' data (ms_witness_create_host_sliced):
        all offsets, the elements [data_first, data_first + len(data_slice)) and the first min(total, 130) elements"""
        remote_heights = remote_heights or {}
        trs = [_u64(t) if t is not None and len(t) else np.zeros((0, 1), dtype=np.uint64) for t in traces]
        n = self.n_circuits
        if len(trs) != n:
            raise MstarkError("expected one trace per circuit")
        tptr = (u64p * n)(*[None if i in remote_heights else _p(t) for i, t in enumerate(trs)])
        hs = _u64([remote_heights.get(i, t.shape[0]) for i, t in enumerate(trs)])
        offs, sl, hd = _u64(claim_offsets), _u64(data_slice), _u64(head)
        sl_arg = sl if sl.size else np.zeros(1, dtype=np.uint64)
        hd_arg = hd if hd.size else np.zeros(1, dtype=np.uint64)
        h = C.c_void_p()
        pinned = C.c_int32(0)
        _check(lib().ms_witness_create_host_sliced(self.h, tptr, _p(hs), C.c_size_t(len(offs) - 1), _p(offs), C.c_uint64(data_first),
                                                   C.c_uint64(sl.size), _p(sl_arg), _p(hd_arg), C.c_size_t(hd.size), C.byref(pinned), C.byref(h)))
        w = SystemWitness(h, int(hs.sum()), self)
        w.keep = trs
        w.pinned = bool(pinned.value)
        return w

    def bench_witness_on_device(self, num_adds, a0=0xDEADBEEF, b0=0xCAFEBABE):
        """The bench workload's witness and claims generated in HBM (benches/multi_stark.rs:171-238) for [ByteTable, U32Add]."""
        h = C.c_void_p()
        _check(lib().ms_witness_u32_add_bench(self.h, C.c_size_t(num_adds), C.c_uint32(a0), C.c_uint32(b0), C.byref(h)))
        height = max(1, 1 << (num_adds - 1).bit_length())
        return SystemWitness(h, 256 + height, self)

    def _out_buffer(self):
        # one output buffer per system, reused: a fresh 2 MB array per proof costs an mmap and a page fault per 4 KB touched
        cap = getattr(self, "_proof_cap", 1 << 21)
        buf = getattr(self, "_proof_buf", None)
        if buf is None or buf.size != cap:
            buf = self._proof_buf = np.zeros(cap, dtype=np.uint8)
        return buf, cap

    def prove_multiple_claims(self, witness, want_times=False):
        times = np.zeros(6, dtype=np.float64)
        while True:
            out, cap = self._out_buffer()
            n = C.c_size_t()
            rc = lib().ms_prove(self.h, witness.h, _b(out), C.c_size_t(cap), C.byref(n),
                                times.ctypes.data_as(C.POINTER(C.c_double)) if want_times else None)
            if rc == -3:
                self._proof_cap = n.value
                continue
            _check(rc)
            keys = ["stage1_commit", "lookup_construction", "stage2_commit", "quotient", "fri_open", "total"]
            return Proof(C.string_at(out.ctypes.data, n.value), dict(zip(keys, times.tolist())) if want_times else None)

    prove = prove_multiple_claims

    def verify_multiple_claims(self, claims_packed, proof) -> int:
        """`System::verify_multiple_claims`: 0 = accepted, else the VerificationError code (2 opening, 3 shape, 4 system,
        5 out-of-domain mismatch, 6 unbalanced channel). `proof`: bytes or a Proof."""
        data = proof.to_bytes() if isinstance(proof, Proof) else bytes(proof)
        buf = np.frombuffer(data, dtype=np.uint8)
        offs, cd = claims_packed
        cd = cd if cd.size else np.zeros(1, dtype=np.uint64)
        verdict = C.c_int32(-1)
        _check(lib().ms_verify(self.h, C.c_size_t(len(offs) - 1), _p(offs), _p(cd), _b(buf), C.c_size_t(len(data)), C.byref(verdict)))
        return int(verdict.value)

    verify = verify_multiple_claims

    def prove_sharded(self, witness, comm, owners, want_times=False):
        """The same Proof computed by `comm.world` ranks (ms_prove_sharded; see multi-stark_amd/sharded.py for `comm`).
        owners[i] = rank computing circuit i, -1 = replicated. Collective: every rank calls it; every rank gets the bytes."""
        times = np.zeros(6, dtype=np.float64)
        own = np.ascontiguousarray(owners, dtype=np.int32)
        if own.size != self.n_circuits:
            raise MstarkError("expected one owner per circuit")
        while True:
            out, cap = self._out_buffer()
            n = C.c_size_t()
            rc = lib().ms_prove_sharded(self.h, witness.h, C.byref(comm.struct), own.ctypes.data_as(C.POINTER(C.c_int32)), _b(out),
                                        C.c_size_t(cap), C.byref(n), times.ctypes.data_as(C.POINTER(C.c_double)) if want_times else None)
            comm.reraise()
            if rc == -3:
                self._proof_cap = n.value
                continue
            _check(rc)
            keys = ["stage1_commit", "lookup_construction", "stage2_commit", "quotient", "fri_open", "total"]
            return Proof(C.string_at(out.ctypes.data, n.value), dict(zip(keys, times.tolist())) if want_times else None)

    def preprocessed_mmcs(self, heights_widths):
        """the ProverKey's preprocessed prover data as a commitment handle (None without preprocessed traces);
        heights_widths: [(lde_height, width)] of the preprocessed matrices"""
        out = C.c_void_p()
        _check(lib().ms_system_preprocessed_mmcs(self.h, C.byref(out)))
        if not out:
            return None
        maxh = max(h for h, _ in heights_widths)
        return DeviceCommitment(self.ctx, out, self.preprocessed_commit(), [w for _, w in heights_widths], maxh.bit_length() - 1)

    def quotient(self, ci, log_n, s1, s1_idx, s2, s2_idx, publics8, alpha):
        """quotient_values + shifted_quotient_slices + lde_from_shifted_coefficients (src/prover.rs:483,511-517) -> LDE handle"""
        out = C.c_void_p()
        _check(lib().ms_quotient(self.h, C.c_size_t(ci), C.c_uint32(log_n), s1.h, C.c_size_t(s1_idx), s2.h, C.c_size_t(s2_idx),
                                 _p(_u64(publics8)), _p(_u64(alpha)), C.byref(out)))
        return Trace(out)

    def quotient_values(self, ci, publics8, log_n, log_q, pre_q, s1_q, s2_q, alpha):
        N = 1 << (log_n + log_q)
        out = np.zeros((N, 2), dtype=np.uint64)
        pre = _u64(pre_q) if pre_q is not None else np.zeros(1, dtype=np.uint64)
        _check(lib().ms_quotient_values(self.h, C.c_size_t(ci), _p(_u64(publics8)), C.c_uint32(log_n), C.c_uint32(log_q), _p(pre),
                                        _p(_u64(s1_q)), _p(_u64(s2_q)), _p(_u64(alpha)), _p(out)))
        return out
