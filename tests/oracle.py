"""ctypes wrapper over oracle/libms_oracle.so (the CPU restatement). Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product.
The same file, executed under the module name `oracle_bb` (tests/oracle_bb.py), binds oracle/libms_oracle_bb.so: the
restatement compiled for the reference's BabyBear / Poseidon2 configuration (extension degree 4)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
BABYBEAR = __name__.endswith("oracle_bb")
# MSO_ORACLE_DIR: another build of the same sources (bench.py's cpu_baseline leg compiles one with -march=native on the box it times)
LIB_PATH = os.path.join(os.environ.get("MSO_ORACLE_DIR") or ORACLE_DIR, "libms_oracle_bb.so" if BABYBEAR else "libms_oracle.so")
D = 4 if BABYBEAR else 2  # extension degree: Ext values cross the C surface as D consecutive u64

# libgomp's default active spinning stalls badly when the container's CPUs are oversubscribed
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

_lib = None
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.mso_last_error.restype = C.c_char_p
        for name in ("mso_gl_mul", "mso_gl_add", "mso_gl_sub"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [C.c_uint64, C.c_uint64]
        L.mso_gl_inv.restype = C.c_uint64
        L.mso_gl_inv.argtypes = [C.c_uint64]
        L.mso_gl_two_adic_generator.restype = C.c_uint64
        L.mso_gl_two_adic_generator.argtypes = [C.c_uint]
        L.mso_mmcs_commit.restype = C.c_void_p
        L.mso_challenger_new.restype = C.c_void_p
        L.mso_system_create.restype = C.c_void_p
        L.mso_prove.restype = C.c_long
        L.mso_challenger_sample_bits.restype = C.c_uint64
        L.mso_challenger_grind.restype = C.c_uint64
        L.mso_challenger_for_params.restype = C.c_void_p
        L.mso_pcs_open.restype = C.c_long
        L.mso_field_order.restype = C.c_uint64
        assert L.mso_ext_degree() == D
        if BABYBEAR:
            L.mso_to_wire.restype = C.c_uint64
            L.mso_to_wire.argtypes = [C.c_uint64]
        _lib = L
    return _lib


def set_threads(n):
    lib().mso_set_threads(int(n))


def max_threads():
    return int(lib().mso_max_threads())


def _err():
    return lib().mso_last_error().decode()


def _p(a):
    return a.ctypes.data_as(u64p)


def _b(a):
    return a.ctypes.data_as(u8p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def set_poseidon2(constants141):
    """BabyBear configuration only: the permutation's round constants (process-wide in the oracle)"""
    k = _u64(constants141).reshape(-1)
    assert k.size == 141
    if lib().mso_set_poseidon2(_p(k)):
        raise RuntimeError("non-canonical round constant")


def poseidon2_permute(state16):
    st = _u64(state16).copy()
    lib().mso_poseidon2_permute(_p(st))
    return st


def hash_bytes(data: bytes) -> bytes:
    out = np.zeros(32, dtype=np.uint8)
    buf = np.frombuffer(data, dtype=np.uint8) if data else np.zeros(0, dtype=np.uint8)
    lib().mso_hash_bytes(_b(buf), C.c_size_t(len(data)), _b(out))
    return out.tobytes()


def hash_elems(elems) -> bytes:
    e = _u64(elems)
    out = np.zeros(32, dtype=np.uint8)
    lib().mso_hash_elems(_p(e), C.c_size_t(e.size), _b(out))
    return out.tobytes()


def compress2(l: bytes, r: bytes) -> bytes:
    out = np.zeros(32, dtype=np.uint8)
    lib().mso_compress2(_b(np.frombuffer(l, dtype=np.uint8)), _b(np.frombuffer(r, dtype=np.uint8)), _b(out))
    return out.tobytes()


def dft_batch(m, inverse=False):
    m = _u64(m)
    out = np.empty_like(m)
    if lib().mso_dft_batch(_p(m), C.c_size_t(m.shape[0]), C.c_size_t(m.shape[1]), int(inverse), _p(out)):
        raise RuntimeError(_err())
    return out


def coset_lde_bitrev(m, log_blowup, shift=31 if BABYBEAR else 7):
    m = _u64(m)
    out = np.empty((m.shape[0] << log_blowup, m.shape[1]), dtype=np.uint64)
    if lib().mso_coset_lde_bitrev(_p(m), C.c_size_t(m.shape[0]), C.c_size_t(m.shape[1]), C.c_uint(log_blowup),
                                  C.c_uint64(shift), _p(out)):
        raise RuntimeError(_err())
    return out


def shifted_quotient_slices(m, qdeg):
    m = _u64(m)
    out = np.empty((m.shape[0] // qdeg, m.shape[1] * qdeg), dtype=np.uint64)
    if lib().mso_shifted_quotient_slices(_p(m), C.c_size_t(m.shape[0]), C.c_size_t(m.shape[1]), C.c_size_t(qdeg), _p(out)):
        raise RuntimeError(_err())
    return out


def lde_from_shifted_coefficients(m, log_blowup):
    m = _u64(m)
    out = np.empty((m.shape[0] << log_blowup, m.shape[1]), dtype=np.uint64)
    if lib().mso_lde_from_shifted_coefficients(_p(m), C.c_size_t(m.shape[0]), C.c_size_t(m.shape[1]),
                                               C.c_uint(log_blowup), _p(out)):
        raise RuntimeError(_err())
    return out


class Mmcs:
    def __init__(self, mats, cap_height=0):
        self.mats = [_u64(m) for m in mats]
        n = len(self.mats)
        ptrs = (u64p * n)(*[_p(m) for m in self.mats])
        hs = _u64([m.shape[0] for m in self.mats])
        ws = _u64([m.shape[1] for m in self.mats])
        cap = np.zeros(32 << cap_height, dtype=np.uint8)
        self.h = lib().mso_mmcs_commit(C.c_size_t(n), ptrs, _p(hs), _p(ws), C.c_uint(cap_height), _b(cap))
        if not self.h:
            raise RuntimeError(_err())
        maxh = int(hs.max())
        self.cap = cap[: 32 * min(1 << cap_height, maxh)].tobytes()
        self.log_max = maxh.bit_length() - 1
        self.heights, self.widths = hs, ws

    def open(self, index):
        vals = np.zeros(int(self.widths.sum()), dtype=np.uint64)
        proof = np.zeros(32 * (self.log_max + 1), dtype=np.uint8)
        k = lib().mso_mmcs_open(C.c_void_p(self.h), C.c_size_t(index), _p(vals), _b(proof))
        if k < 0:
            raise RuntimeError(_err())
        return vals, proof[: 32 * k].tobytes()

    def verify(self, index, vals, proof, cap=None):
        cap = self.cap if cap is None else cap
        capa = np.frombuffer(cap, dtype=np.uint8)
        pa = np.frombuffer(proof, dtype=np.uint8) if proof else np.zeros(0, dtype=np.uint8)
        vals = _u64(vals)
        return lib().mso_mmcs_verify(_b(capa), C.c_size_t(len(cap) // 32), C.c_size_t(len(self.mats)), _p(self.heights),
                                     _p(self.widths), C.c_size_t(index), _p(vals), _b(pa), C.c_size_t(len(proof) // 32))

    def __del__(self):
        if getattr(self, "h", None):
            lib().mso_mmcs_free(C.c_void_p(self.h))


class Challenger:
    def __init__(self, seed: bytes = b""):
        s = np.frombuffer(seed, dtype=np.uint8) if seed else np.zeros(0, dtype=np.uint8)
        self.h = lib().mso_challenger_new(_b(s), C.c_size_t(len(seed)))

    @staticmethod
    def for_params(params):
        """config.initialise_challenger() (src/types.rs:118-130)"""
        c = Challenger.__new__(Challenger)
        c.h = lib().mso_challenger_for_params(_p(_u64(params.words())))
        return c

    def observe(self, x):
        lib().mso_challenger_observe(C.c_void_p(self.h), C.c_uint64(x))

    def observe_digests(self, cap: bytes):
        a = np.frombuffer(cap, dtype=np.uint8)
        lib().mso_challenger_observe_digests(C.c_void_p(self.h), _b(a), C.c_size_t(len(cap) // 32))

    def observe_bytes(self, b: bytes):
        a = np.frombuffer(b, dtype=np.uint8)
        lib().mso_challenger_observe_bytes(C.c_void_p(self.h), _b(a), C.c_size_t(len(b)))

    def sample_ext(self):
        o = np.zeros(D, dtype=np.uint64)
        lib().mso_challenger_sample_ext(C.c_void_p(self.h), _p(o))
        return tuple(int(x) for x in o)

    def sample_bits(self, bits):
        return int(lib().mso_challenger_sample_bits(C.c_void_p(self.h), C.c_uint(bits)))

    def grind(self, bits):
        return int(lib().mso_challenger_grind(C.c_void_p(self.h), C.c_uint(bits)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().mso_challenger_free(C.c_void_p(self.h))


def _flatten_points(rounds):
    n_points, pts = [], []
    for _mmcs, per_matrix in rounds:
        for plist in per_matrix:
            n_points.append(len(plist))
            for z in plist:
                pts.extend(int(x) for x in z)
    return _u64(n_points), _u64(pts if pts else [0])


def pcs_open(params, rounds, challenger):
    """Pcs::open (examples/pcs_example.rs:88-95, src/prover.rs:580). rounds: [(Mmcs, [[point, ..] per matrix])], a point
    = D ints. Returns (opened values flat: round -> matrix -> point -> column, D words each; FriProof bytes)."""
    n_points, pts = _flatten_points(rounds)
    handles = (C.c_void_p * len(rounds))(*[C.c_void_p(m.h) for m, _ in rounds])
    total = sum(len(plist) * int(m.widths[i]) for m, per in rounds for i, plist in enumerate(per))
    opened = np.zeros(max(total, 1) * D, dtype=np.uint64)
    cap = 1 << 22
    while True:
        out = np.zeros(cap, dtype=np.uint8)
        r = lib().mso_pcs_open(_p(_u64(params.words())), C.c_size_t(len(rounds)), handles, _p(n_points), _p(pts), C.c_void_p(challenger.h),
                               _p(opened), _b(out), C.c_size_t(cap))
        if r >= 0:
            return opened[: total * D].copy(), out[:r].tobytes()
        if r == -1:
            raise RuntimeError(_err())
        cap = -r


def pcs_verify(params, rounds, opened, fri: bytes, challenger):
    """Pcs::verify. rounds: [(cap bytes, [(log_n, width)] per matrix, [[point, ..] per matrix])]; True = accepted"""
    caps = [np.frombuffer(c, dtype=np.uint8) for c, _, _ in rounds]
    cap_ptrs = (u8p * len(rounds))(*[_b(c) for c in caps])
    cap_sizes = _u64([len(c) // 32 for c, _, _ in rounds])
    n_mats = _u64([len(d) for _, d, _ in rounds])
    log_n = _u64([ln for _, d, _ in rounds for ln, _w in d])
    widths = _u64([w for _, d, _ in rounds for _ln, w in d])
    n_points, pts = _flatten_points([(None, per) for _, _, per in rounds])
    f = np.frombuffer(fri, dtype=np.uint8) if fri else np.zeros(1, dtype=np.uint8)
    op = _u64(opened) if len(opened) else np.zeros(1, dtype=np.uint64)
    r = lib().mso_pcs_verify(_p(_u64(params.words())), C.c_size_t(len(rounds)), cap_ptrs, _p(cap_sizes), _p(n_mats), _p(log_n), _p(widths),
                             _p(n_points), _p(pts), _p(op), _b(f), C.c_size_t(len(fri)), C.c_void_p(challenger.h))
    if r < 0:
        raise RuntimeError(_err())
    return r == 1


class System:
    """Oracle-side System + ProverKey built from a front-end blob."""

    def __init__(self, blob: bytes):
        a = np.frombuffer(blob, dtype=np.uint8)
        self.h = lib().mso_system_create(_b(a), C.c_size_t(len(blob)))
        if not self.h:
            raise RuntimeError(_err())

    def __del__(self):
        if getattr(self, "h", None):
            lib().mso_system_free(C.c_void_p(self.h))

    def preprocessed_commit(self):
        out = np.zeros(32 * 256, dtype=np.uint8)
        k = lib().mso_system_preprocessed_commit(C.c_void_p(self.h), _b(out))
        return out[: 32 * k].tobytes() if k else None

    def circuit_info(self, ci):
        o = np.zeros(9, dtype=np.uint64)
        if lib().mso_system_circuit_info(C.c_void_p(self.h), C.c_size_t(ci), _p(o)):
            raise RuntimeError(_err())
        keys = ["main_width", "pre_width", "pre_height", "num_lookups", "stage2_width", "constraint_count",
                "max_constraint_degree", "quotient_degree", "args_width"]
        return dict(zip(keys, (int(x) for x in o)))

    def compute_lookup_values(self, ci, trace):
        info = self.circuit_info(ci)
        tr = _u64(trace)
        h = tr.shape[0]
        mult = np.zeros((h, info["num_lookups"]), dtype=np.uint64)
        args = np.zeros((h, info["args_width"]), dtype=np.uint64)
        if lib().mso_compute_lookup_values(C.c_void_p(self.h), C.c_size_t(ci), _p(tr), C.c_size_t(h), _p(mult), _p(args)):
            raise RuntimeError(_err())
        return mult, args

    def prove(self, traces, claims_packed, want_times=False):
        offs, data = claims_packed
        trs = [_u64(t) for t in traces]
        n = len(trs)
        ptrs = (u64p * n)(*[_p(t) for t in trs])
        hs = _u64([t.shape[0] for t in trs])
        times = np.zeros(6, dtype=np.float64)
        cap = 1 << 22
        while True:
            out = np.zeros(cap, dtype=np.uint8)
            r = lib().mso_prove(C.c_void_p(self.h), C.c_size_t(len(offs) - 1), _p(offs), _p(data), ptrs, _p(hs), _b(out),
                                C.c_size_t(cap), times.ctypes.data_as(C.POINTER(C.c_double)))
            if r >= 0:
                proof = out[:r].tobytes()
                break
            if r == -1:
                raise RuntimeError(_err())
            cap = -r
        if want_times:
            keys = ["stage1_commit", "lookup_construction", "stage2_commit", "quotient", "fri_open", "total"]
            return proof, dict(zip(keys, times.tolist()))
        return proof

    def verify(self, claims_packed, proof: bytes):
        offs, data = claims_packed
        a = np.frombuffer(proof, dtype=np.uint8)
        return lib().mso_verify(C.c_void_p(self.h), C.c_size_t(len(offs) - 1), _p(offs), _p(data), _b(a), C.c_size_t(len(proof)))


def stage2_trace(mult, arg_offsets, args, beta, gamma, acc_in):
    mult, args, arg_offsets = _u64(mult), _u64(args), _u64(arg_offsets)
    h, L = mult.shape
    tr = np.zeros((h, max(L, 1) * D), dtype=np.uint64)
    acc = np.zeros(D, dtype=np.uint64)
    if lib().mso_stage2_trace(C.c_size_t(h), C.c_size_t(L), _p(mult), _p(arg_offsets), _p(args), _p(_u64(beta)),
                              _p(_u64(gamma)), _p(_u64(acc_in)), _p(tr), _p(acc)):
        raise RuntimeError(_err())
    return tr, tuple(int(x) for x in acc)


def claims_accumulator(claims_packed, beta, gamma):
    offs, data = claims_packed
    acc = np.zeros(D, dtype=np.uint64)
    if lib().mso_claims_accumulator(C.c_size_t(len(offs) - 1), _p(offs), _p(data), _p(_u64(beta)), _p(_u64(gamma)), _p(acc)):
        raise RuntimeError(_err())
    return tuple(int(x) for x in acc)


def quotient_values(system, ci, publics8, log_n, log_q, pre_q, s1_q, s2_q, alpha):
    N = 1 << (log_n + log_q)
    out = np.zeros((N, D), dtype=np.uint64)
    pre = _u64(pre_q) if pre_q is not None else np.zeros(1, dtype=np.uint64)
    if lib().mso_quotient_values(C.c_void_p(system.h), C.c_size_t(ci), _p(_u64(publics8)), C.c_uint(log_n), C.c_uint(log_q),
                                 _p(pre), _p(_u64(s1_q)), _p(_u64(s2_q)), _p(_u64(alpha)), _p(out)):
        raise RuntimeError(_err())
    return out


def selectors_on_coset(log_n, log_q):
    N = 1 << (log_n + log_q)
    outs = [np.zeros(N, dtype=np.uint64) for _ in range(4)]
    if lib().mso_selectors_on_coset(C.c_uint(log_n), C.c_uint(log_q), *[_p(o) for o in outs]):
        raise RuntimeError(_err())
    return outs
