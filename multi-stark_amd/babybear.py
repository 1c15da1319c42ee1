"""The reference's second configuration (BabyBear, degree-4 extension, Poseidon2, DuplexChallenger -
/root/reference/src/test_circuits/baby_bear_config.rs) over the C ABI of include/mstark_bb.h. Same interface as the
Goldilocks classes of the package: `System.new(ctx, params, circuits, poseidon2)`, `system.witness(traces, claims)`,
`system.prove_multiple_claims(witness)` -> `Proof.to_bytes()`. Elements cross the ABI as canonical u32.
All computation happens in libmstark_hip.so; there is no CPU fallback."""
import ctypes as C

import numpy as np

from . import frontend

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
P = frontend.BABYBEAR["P"]


def exported_symbols():
    """Every entry point include/mstark_bb.h declares (used by the CPU-side ABI test)."""
    return ["msbb_system_create", "msbb_system_destroy", "msbb_system_preprocessed_commit", "msbb_system_circuit_info",
            "msbb_witness_create", "msbb_witness_create_host", "msbb_witness_destroy", "msbb_prove", "msbb_verify", "msbb_set_poseidon2", "msbb_poseidon2_permute",
            "msbb_dft_batch", "msbb_coset_lde_batch", "msbb_mmcs_commit", "msbb_mmcs_open", "msbb_mmcs_destroy", "msbb_field_op",
            "msbb_challenger_create", "msbb_challenger_destroy", "msbb_challenger_observe", "msbb_challenger_observe_digests",
            "msbb_challenger_sample_ext", "msbb_challenger_sample_bits", "msbb_challenger_observe_claims", "msbb_trace_destroy", "msbb_trace_info",
            "msbb_system_preprocessed_mmcs", "msbb_witness_commit_stage1", "msbb_witness_claims_accumulator", "msbb_stage2_build",
            "msbb_pcs_commit_traces", "msbb_quotient", "msbb_pcs_commit_ldes", "msbb_pcs_open"]


import sys

_PKG = sys.modules[__name__.rsplit(".", 1)[0]]  # the parent package is being imported when this module loads


def _pkg():
    return _PKG


def _lib():
    return _pkg().lib()


def _check(rc):
    if rc != 0:
        raise _pkg().MstarkError(_lib().ms_last_error().decode() or ("mstark error %d" % rc))


def _u32(a):
    a = np.asarray(a)
    if a.size and int(a.max()) >= P:
        raise _pkg().MstarkError("non-canonical BabyBear element in input")
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p32(a):
    return a.ctypes.data_as(u32p)


class Proof:
    def __init__(self, data: bytes, stage_ms=None):
        self._data, self.stage_ms = data, stage_ms

    def to_bytes(self):
        return self._data


class Witness:
    def __init__(self, system, traces, claims_packed, host_resident=False):
        """host_resident: the witness stays in host memory (msbb_witness_create_host) and every proof uploads it - the
        reference's timed region; the arrays are kept alive (and page-locked) by this object"""
        self.system = system
        offs, data = claims_packed
        trs = [_u32(t) for t in traces]
        n = len(trs)
        ptrs = (u32p * n)(*[_p32(t) for t in trs])
        hs = np.ascontiguousarray([t.shape[0] for t in trs], dtype=np.uint64)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        data = _u32(data)
        self.h = C.c_void_p()
        if host_resident:
            pinned = C.c_int32(0)
            _check(_lib().msbb_witness_create_host(system.h, ptrs, hs.ctypes.data_as(u64p), C.c_size_t(len(offs) - 1), offs.ctypes.data_as(u64p),
                                                   _p32(data), C.byref(pinned), C.byref(self.h)))
            self.keep, self.pinned = trs, bool(pinned.value)
        else:
            _check(_lib().msbb_witness_create(system.h, ptrs, hs.ctypes.data_as(u64p), C.c_size_t(len(offs) - 1), offs.ctypes.data_as(u64p),
                                              _p32(data), C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None):
            _lib().msbb_witness_destroy(self.h)
            self.h = None


class System:
    """System<BabyBearPoseidon2Config> + ProverKey (msbb_system)."""

    def __init__(self, ctx, blob: bytes, n_circuits):
        self.ctx, self.blob, self.n_circuits = ctx, blob, n_circuits
        a = np.frombuffer(blob, dtype=np.uint8)
        self.h = C.c_void_p()
        _check(_lib().msbb_system_create(ctx.h, a.ctypes.data_as(u8p), C.c_size_t(len(blob)), C.byref(self.h)))

    @staticmethod
    def new(ctx, params, circuit_inputs, poseidon2):
        with frontend.field(frontend.BABYBEAR):
            compiled = [frontend.compile_circuit(c) for c in circuit_inputs]
            blob = frontend.system_blob(params, compiled, poseidon2)
        return System(ctx, blob, len(compiled))

    def __del__(self):
        if getattr(self, "h", None):
            _lib().msbb_system_destroy(self.h)
            self.h = None

    def circuit_info(self, ci):
        o = np.zeros(9, dtype=np.uint64)
        _check(_lib().msbb_system_circuit_info(self.h, C.c_size_t(ci), o.ctypes.data_as(u64p)))
        keys = ["main_width", "pre_width", "pre_height", "num_lookups", "stage2_width", "constraint_count", "max_constraint_degree",
                "quotient_degree", "args_width"]
        return dict(zip(keys, (int(x) for x in o)))

    def preprocessed_commit(self):
        out = np.zeros(8 * 256, dtype=np.uint32)
        n = C.c_size_t()
        _check(_lib().msbb_system_preprocessed_commit(self.h, _p32(out), C.c_size_t(out.size), C.byref(n)))
        return out[: 8 * n.value].copy() if n.value else None

    def witness(self, traces, claims_packed):
        return Witness(self, traces, claims_packed)

    def host_witness(self, traces, claims_packed):
        """a SystemWitness that stays in host memory: every prove_multiple_claims uploads it (msbb_witness_create_host)"""
        return Witness(self, traces, claims_packed, host_resident=True)

    def verify_multiple_claims(self, claims_packed, proof: bytes):
        """0 = accepted, otherwise the reference's VerificationError variant (MS_VERDICT_*)"""
        offs, data = claims_packed
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        data = np.ascontiguousarray(data, dtype=np.uint32)
        a = np.frombuffer(proof, dtype=np.uint8) if proof else np.zeros(1, dtype=np.uint8)
        verdict = C.c_int32()
        _check(_lib().msbb_verify(self.h, C.c_size_t(len(offs) - 1), offs.ctypes.data_as(u64p), _p32(data), a.ctypes.data_as(u8p),
                                  C.c_size_t(len(proof)), C.byref(verdict)))
        return verdict.value

    verify = verify_multiple_claims

    def prove_multiple_claims(self, witness, want_times=False):
        times = np.zeros(6, dtype=np.float64)
        while True:
            # one output buffer per system, reused: a fresh multi-megabyte array per proof costs an mmap and its page faults
            cap = getattr(self, "_proof_cap", 1 << 22)
            out = getattr(self, "_proof_buf", None)
            if out is None or out.size != cap:
                out = self._proof_buf = np.zeros(cap, dtype=np.uint8)
            n = C.c_size_t()
            rc = _lib().msbb_prove(self.h, witness.h, out.ctypes.data_as(u8p), C.c_size_t(cap), C.byref(n),
                                   times.ctypes.data_as(C.POINTER(C.c_double)) if want_times else None)
            if rc == -3:
                self._proof_cap = n.value
                continue
            _check(rc)
            keys = ["stage1_commit", "lookup_construction", "stage2_commit", "quotient", "fri_open", "total"]
            return Proof(out[: n.value].tobytes(), dict(zip(keys, times.tolist())) if want_times else None)


# ---- Level 2: the prover's steps on device handles (include/mstark_bb.h; tests/test_gpu_bb_level2.py drives the reference's loop)
def _e4(x):
    a = _u32(x).reshape(-1)
    assert a.size == 4
    return a


class Challenger:
    """config.initialise_challenger() of the system's configuration (msbb_challenger_*): values in and out are canonical"""

    def __init__(self, system):
        self.system = system
        self.h = C.c_void_p()
        _check(_lib().msbb_challenger_create(system.h, C.byref(self.h)))

    def observe(self, elems):
        a = _u32(np.array([int(x) % P for x in elems], dtype=np.uint64))   # (Val::from_usize for the integers of the shape)
        _check(_lib().msbb_challenger_observe(self.h, _p32(a), C.c_size_t(a.size)))

    def observe_digests(self, words):
        a = _u32(words).reshape(-1)
        assert a.size % 8 == 0
        _check(_lib().msbb_challenger_observe_digests(self.h, _p32(a), C.c_size_t(a.size // 8)))

    def observe_claims(self, witness):
        _check(_lib().msbb_challenger_observe_claims(self.h, witness.h))

    def sample_ext(self):
        o = np.zeros(4, dtype=np.uint32)
        _check(_lib().msbb_challenger_sample_ext(self.h, _p32(o)))
        return tuple(int(x) for x in o)

    def sample_bits(self, bits):
        o = C.c_uint64()
        _check(_lib().msbb_challenger_sample_bits(self.h, C.c_uint32(bits), C.byref(o)))
        return o.value

    def __del__(self):
        if getattr(self, "h", None):
            _lib().msbb_challenger_destroy(self.h)
            self.h = None


class Trace:
    """a matrix that stays in HBM between two steps (msbb_trace)"""

    def __init__(self, h):
        self.h = h

    def info(self):
        o = np.zeros(3, dtype=np.uint64)
        _check(_lib().msbb_trace_info(self.h, o.ctypes.data_as(u64p)))
        return tuple(int(x) for x in o)

    def __del__(self):
        if getattr(self, "h", None):
            _lib().msbb_trace_destroy(self.h)
            self.h = None


class Committed:
    """prover data of one commitment (msbb_mmcs) with its cap"""

    def __init__(self, h, cap, shapes=None):
        self.h, self.cap, self.shapes = h, cap, shapes

    def __del__(self):
        if getattr(self, "h", None):
            _lib().msbb_mmcs_destroy(self.h)
            self.h = None


def _cap_buf(cap_height):
    return np.zeros(8 << cap_height, dtype=np.uint32)


def _trim_cap(cap, cap_height, max_height):
    return cap[: 8 * min(1 << cap_height, max_height)].copy()


def commit_stage1(witness, params, lde_heights):
    cap = _cap_buf(params.cap_height)
    h = C.c_void_p()
    _check(_lib().msbb_witness_commit_stage1(witness.h, _p32(cap), C.byref(h)))
    return Committed(h, _trim_cap(cap, params.cap_height, max(lde_heights)))


def claims_accumulator(witness, beta, gamma):
    o = np.zeros(4, dtype=np.uint32)
    _check(_lib().msbb_witness_claims_accumulator(witness.h, _p32(_e4(beta)), _p32(_e4(gamma)), _p32(o)))
    return tuple(int(x) for x in o)


def stage2_build(witness, n_active, beta, gamma, acc_in):
    accs = np.zeros(4 * n_active, dtype=np.uint32)
    hs = (C.c_void_p * n_active)()
    _check(_lib().msbb_stage2_build(witness.h, _p32(_e4(beta)), _p32(_e4(gamma)), _p32(_e4(acc_in)), _p32(accs), hs))
    return [tuple(int(x) for x in accs[4 * i: 4 * i + 4]) for i in range(n_active)], [Trace(C.c_void_p(h)) for h in hs]


def _commit(fn, system, params, traces, lde_heights):
    n = len(traces)
    hs = (C.c_void_p * n)(*[t.h for t in traces])
    cap = _cap_buf(params.cap_height)
    h = C.c_void_p()
    _check(fn(system.h, C.c_size_t(n), hs, _p32(cap), C.byref(h)))
    return Committed(h, _trim_cap(cap, params.cap_height, max(lde_heights)))


def pcs_commit_traces(system, params, traces):
    heights = [t.info()[0] << params.log_blowup for t in traces]
    return _commit(_lib().msbb_pcs_commit_traces, system, params, traces, heights)


def pcs_commit_ldes(system, params, ldes):
    heights = [t.info()[0] for t in ldes]
    return _commit(_lib().msbb_pcs_commit_ldes, system, params, ldes, heights)


def quotient(system, circuit, log_n, s1, s1_idx, s2, s2_idx, publics16, alpha):
    pub = _u32(publics16).reshape(-1)
    assert pub.size == 16
    h = C.c_void_p()
    _check(_lib().msbb_quotient(system.h, C.c_size_t(circuit), C.c_uint32(log_n), s1.h, C.c_size_t(s1_idx), s2.h, C.c_size_t(s2_idx), _p32(pub),
                                _p32(_e4(alpha)), C.byref(h)))
    return Trace(h)


def preprocessed_mmcs(system):
    h = C.c_void_p()
    _check(_lib().msbb_system_preprocessed_mmcs(system.h, C.byref(h)))
    return Committed(h, system.preprocessed_commit()) if h.value else None


def pcs_open(system, rounds, challenger):
    """rounds: [(Committed, [widths], [[point, ...] per matrix])]; returns (opened canonical words, FriProof bytes)"""
    hs = (C.c_void_p * len(rounds))(*[r[0].h for r in rounds])
    npts, pts, words = [], [], 0
    for _, widths, points in rounds:
        assert len(widths) == len(points)
        for w, ps in zip(widths, points):
            npts.append(len(ps))
            for p_ in ps:
                pts.extend(int(x) for x in p_)
            words += 4 * w * len(ps)
    npts = np.ascontiguousarray(npts, dtype=np.uint64)
    pts = _u32(pts if pts else [0])
    opened = np.zeros(max(words, 1), dtype=np.uint32)
    cap = 1 << 20
    while True:
        fri = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t()
        rc = _lib().msbb_pcs_open(system.h, C.c_size_t(len(rounds)), hs, npts.ctypes.data_as(u64p), _p32(pts), challenger.h, _p32(opened),
                                  C.c_size_t(opened.size), fri.ctypes.data_as(u8p), C.c_size_t(cap), C.byref(n))
        if rc == -3 and n.value > cap:
            raise _pkg().MstarkError("msbb_pcs_open: FriProof larger than the buffer (%d bytes); the challenger has been advanced" % n.value)
        _check(rc)
        return opened[:words].copy(), fri[: n.value].tobytes()


# ---- PCS-level entry points
def set_poseidon2(ctx, constants141):
    k = _u32(constants141).reshape(-1)
    assert k.size == 141
    _check(_lib().msbb_set_poseidon2(ctx.h, _p32(k)))


def poseidon2_permute(ctx, states):
    st = _u32(states).reshape(-1, 16).copy()
    _check(_lib().msbb_poseidon2_permute(ctx.h, _p32(st), C.c_size_t(st.shape[0])))
    return st


def dft_batch(ctx, m, inverse=False):
    m = _u32(m)
    out = np.empty_like(m)
    _check(_lib().msbb_dft_batch(ctx.h, _p32(m), C.c_size_t(m.shape[0]), C.c_size_t(m.shape[1]), C.c_int32(int(inverse)), _p32(out)))
    return out


def coset_lde_batch(ctx, m, log_blowup):
    m = _u32(m)
    out = np.empty((m.shape[0] << log_blowup, m.shape[1]), dtype=np.uint32)
    _check(_lib().msbb_coset_lde_batch(ctx.h, _p32(m), C.c_size_t(m.shape[0]), C.c_size_t(m.shape[1]), C.c_uint32(log_blowup), _p32(out)))
    return out


class Mmcs:
    def __init__(self, ctx, mats, cap_height=0):
        self.ctx = ctx
        self.mats = [_u32(m) for m in mats]
        n = len(self.mats)
        ptrs = (u32p * n)(*[_p32(m) for m in self.mats])
        hs = np.ascontiguousarray([m.shape[0] for m in self.mats], dtype=np.uint64)
        ws = np.ascontiguousarray([m.shape[1] for m in self.mats], dtype=np.uint64)
        maxh = int(hs.max())
        cap = np.zeros(8 << cap_height, dtype=np.uint32)
        self.h = C.c_void_p()
        _check(_lib().msbb_mmcs_commit(ctx.h, C.c_size_t(n), ptrs, hs.ctypes.data_as(u64p), ws.ctypes.data_as(u64p), C.c_uint32(cap_height),
                                       _p32(cap), C.byref(self.h)))
        self.cap = cap[: 8 * min(1 << cap_height, maxh)].copy()
        self.widths, self.log_max = ws, maxh.bit_length() - 1

    def open(self, index):
        vals = np.zeros(int(self.widths.sum()), dtype=np.uint32)
        proof = np.zeros(8 * (self.log_max + 1), dtype=np.uint32)
        k = C.c_size_t()
        _check(_lib().msbb_mmcs_open(self.h, C.c_size_t(index), _p32(vals), _p32(proof), C.byref(k)))
        return vals, proof[: 8 * k.value].copy()

    def __del__(self):
        if getattr(self, "h", None):
            _lib().msbb_mmcs_destroy(self.h)
            self.h = None


def field_op(ctx, op, a, b=None):
    a = _u32(a)
    out = np.empty_like(a)
    n = a.size // 4 if op >= 4 else a.size
    bb = _u32(b) if b is not None else None
    _check(_lib().msbb_field_op(ctx.h, C.c_int32(op), _p32(a), _p32(bb) if bb is not None else None, C.c_size_t(n), _p32(out)))
    return out
