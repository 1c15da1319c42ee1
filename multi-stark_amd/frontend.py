"""Circuit front-end: expression trees -> compiled node program -> system blob.

Host-side mirror of the reference's authoring path so that the node programs fed to the prover have the
same shape the reference would compile:
  * `Expr` operators with constant folding     -- /root/reference/src/expr.rs:179-227
  * hash-consing interner + `compile`            -- /root/reference/src/graph.rs:120-188, 213-507
  * recording AIR builder (`assert_zero`, `when`, `assert_bool`, ...) -- /root/reference/src/p3_adapter.rs:245-354
    (p3-air / p3-field default methods restated from Plonky3 0.5.1: assert_eq(x,y) = assert_zero(x-y),
     assert_bool(x) = assert_zero(x.bool_check()) with bool_check(x) = (1-x)*x, when(c).assert_zero(x) = c*x)
  * the bench / test circuits and their witnesses -- benches/multi_stark.rs:73-238, examples/simple_proof.rs,
    src/lookup.rs:868-1007 (even/odd), src/test_circuits/u32_add.rs:150-221

This is setup-time code (untimed in the reference's bench): pure Python + numpy, no GPU.
"""
import struct

import numpy as np

import contextlib

P = (1 << 64) - (1 << 32) + 1
EXT_D, EXT_W = 2, 7  # Challenge = BinomialExtensionField<Goldilocks, 2>, X^2 = 7

# node kinds (the system-blob encoding, see include/mstark.h)
N_CONST, N_VAR, N_PUBLIC, N_IS_FIRST, N_IS_LAST, N_IS_TRANS, N_ADD, N_SUB, N_MUL, N_NEG = range(10)
SRC_PRE, SRC_MAIN, SRC_STAGE2 = 0, 1, 2
BLOB_MAGIC = 0x31305359534D0000

# The two StarkGenericConfig instantiations of the reference: GoldilocksBlake3Config (src/types.rs:24-29,199-223) and
# the BabyBear / degree-4 extension / Poseidon2 test configuration (src/test_circuits/baby_bear_config.rs:28-38).
GOLDILOCKS = dict(name="goldilocks", P=P, EXT_D=2, EXT_W=7, BLOB_MAGIC=0x31305359534D0000)
BABYBEAR = dict(name="babybear", P=(1 << 31) - (1 << 27) + 1, EXT_D=4, EXT_W=11, BLOB_MAGIC=0x31304259534D0000)


@contextlib.contextmanager
def field(cfg):
    """Author circuits / witnesses / blobs over another configuration's field: `with frontend.field(frontend.BABYBEAR):`.
    (The reference picks the field through the SC type parameter; this front-end keeps it in module state.)"""
    g = globals()
    saved = {k: g[k] for k in ("P", "EXT_D", "EXT_W", "BLOB_MAGIC")}
    g.update({k: cfg[k] for k in saved})
    try:
        yield cfg
    finally:
        g.update(saved)


class Expr:
    """Frontend base-field expression (src/expr.rs:38-51)."""

    __slots__ = ("kind", "a", "b", "source", "offset")

    def __init__(self, kind, a=None, b=None, source=0, offset=0):
        self.kind, self.a, self.b, self.source, self.offset = kind, a, b, source, offset

    # constructors
    @staticmethod
    def const(v):
        return Expr(N_CONST, int(v) % P)

    @staticmethod
    def var(source, offset, index):
        return Expr(N_VAR, int(index), None, source, offset)

    @staticmethod
    def main(i):
        return Expr.var(SRC_MAIN, 0, i)

    @staticmethod
    def main_next(i):
        return Expr.var(SRC_MAIN, 1, i)

    @staticmethod
    def preprocessed(i):
        return Expr.var(SRC_PRE, 0, i)

    @staticmethod
    def public(i):
        return Expr(N_PUBLIC, int(i))

    def is_const(self, v=None):
        return self.kind == N_CONST and (v is None or self.a == v % P)

    @staticmethod
    def lift(x):
        return x if isinstance(x, Expr) else Expr.const(x)

    # operators, folding constants exactly like src/expr.rs:179-227
    def __add__(self, rhs):
        rhs = Expr.lift(rhs)
        if self.kind == N_CONST and rhs.kind == N_CONST:
            return Expr.const(self.a + rhs.a)
        if self.is_const(0):
            return rhs
        if rhs.is_const(0):
            return self
        return Expr(N_ADD, self, rhs)

    def __radd__(self, lhs):
        return Expr.lift(lhs) + self

    def __sub__(self, rhs):
        rhs = Expr.lift(rhs)
        if self.kind == N_CONST and rhs.kind == N_CONST:
            return Expr.const(self.a - rhs.a)
        if rhs.is_const(0):
            return self
        if self.is_const(0):
            return -rhs
        return Expr(N_SUB, self, rhs)

    def __rsub__(self, lhs):
        return Expr.lift(lhs) - self

    def __mul__(self, rhs):
        rhs = Expr.lift(rhs)
        if self.kind == N_CONST and rhs.kind == N_CONST:
            return Expr.const(self.a * rhs.a)
        if self.is_const(0) or rhs.is_const(0):
            return Expr.const(0)
        if self.is_const(1):
            return rhs
        if rhs.is_const(1):
            return self
        return Expr(N_MUL, self, rhs)

    def __rmul__(self, lhs):
        return Expr.lift(lhs) * self

    def __neg__(self):
        if self.kind == N_CONST:
            return Expr.const(-self.a)
        if self.kind == N_NEG:
            return self.a
        return Expr(N_NEG, self)


IS_FIRST_ROW = Expr(N_IS_FIRST)
IS_LAST_ROW = Expr(N_IS_LAST)
IS_TRANSITION = Expr(N_IS_TRANS)


class ExtExpr:
    """Frontend extension expression (src/expr.rs:55-66): kinds 'coords' | 'base' | 'add' | 'sub' | 'mul' | 'neg'."""

    def __init__(self, kind, a=None, b=None):
        self.kind, self.a, self.b = kind, a, b

    @staticmethod
    def coords(cs):
        return ExtExpr("coords", [Expr.lift(c) for c in cs])

    @staticmethod
    def base(e):
        return ExtExpr("base", Expr.lift(e))

    @staticmethod
    def lift(x):
        return x if isinstance(x, ExtExpr) else ExtExpr.base(x)

    def __add__(self, r):
        return ExtExpr("add", self, ExtExpr.lift(r))

    def __sub__(self, r):
        return ExtExpr("sub", self, ExtExpr.lift(r))

    def __mul__(self, r):
        return ExtExpr("mul", self, ExtExpr.lift(r))

    def __neg__(self):
        return ExtExpr("neg", self)

    def is_purely_base(self):
        if self.kind == "coords":
            return False
        if self.kind == "base":
            return True
        if self.kind == "neg":
            return self.a.is_purely_base()
        return self.a.is_purely_base() and self.b.is_purely_base()


class Lookup:
    """src/lookup.rs:38-74."""

    def __init__(self, multiplicity, args):
        self.multiplicity = Expr.lift(multiplicity)
        self.args = [Expr.lift(a) for a in args]

    @staticmethod
    def push(m, args):
        return Lookup(m, args)

    @staticmethod
    def pull(m, args):
        return Lookup(-Expr.lift(m), args)


class CompileError(Exception):
    pass


class _Interner:
    """Bottom-up hash-consing interner, src/graph.rs:213-324."""

    def __init__(self):
        self.nodes = []  # tuples (kind, source, offset, a, b)
        self.map = {}

    def intern(self, node):
        i = self.map.get(node)
        if i is None:
            i = len(self.nodes)
            self.nodes.append(node)
            self.map[node] = i
        return i

    def as_const(self, i):
        n = self.nodes[i]
        return n[3] if n[0] == N_CONST else None

    def constant(self, v):
        return self.intern((N_CONST, 0, 0, v % P, 0))

    def add(self, a, b):
        x, y = self.as_const(a), self.as_const(b)
        if x is not None and y is not None:
            return self.constant(x + y)
        if x == 0:
            return b
        if y == 0:
            return a
        if a > b:
            a, b = b, a
        return self.intern((N_ADD, 0, 0, a, b))

    def sub(self, a, b):
        if a == b:
            return self.constant(0)
        x, y = self.as_const(a), self.as_const(b)
        if x is not None and y is not None:
            return self.constant(x - y)
        if y == 0:
            return a
        if x == 0:
            return self.neg(b)
        return self.intern((N_SUB, 0, 0, a, b))

    def mul(self, a, b):
        x, y = self.as_const(a), self.as_const(b)
        if x is not None and y is not None:
            return self.constant(x * y)
        if x is not None:
            if x == 0:
                return a
            if x == 1:
                return b
        if y is not None:
            if y == 0:
                return b
            if y == 1:
                return a
        if a > b:
            a, b = b, a
        return self.intern((N_MUL, 0, 0, a, b))

    def neg(self, a):
        x = self.as_const(a)
        if x is not None:
            return self.constant(-x)
        n = self.nodes[a]
        if n[0] == N_NEG:
            return n[3]
        return self.intern((N_NEG, 0, 0, a, 0))

    def compile_expr(self, e, spec, allow_stage2):
        k = e.kind
        if k == N_CONST:
            return self.constant(e.a)
        if k == N_VAR:
            width = {SRC_PRE: spec["preprocessed_width"], SRC_MAIN: spec["main_width"], SRC_STAGE2: spec["stage2_width"]}[e.source]
            if e.source == SRC_STAGE2 and not allow_stage2:
                raise CompileError("Stage2InBaseContext")
            if e.a >= width:
                raise CompileError("ColumnOutOfRange source=%d index=%d width=%d" % (e.source, e.a, width))
            return self.intern((N_VAR, e.source, e.offset, e.a, 0))
        if k == N_PUBLIC:
            if e.a >= spec["num_publics"]:
                raise CompileError("PublicOutOfRange")
            return self.intern((N_PUBLIC, 0, 0, e.a, 0))
        if k in (N_IS_FIRST, N_IS_LAST, N_IS_TRANS):
            return self.intern((k, 0, 0, 0, 0))
        if k == N_NEG:
            return self.neg(self.compile_expr(e.a, spec, allow_stage2))
        a = self.compile_expr(e.a, spec, allow_stage2)
        b = self.compile_expr(e.b, spec, allow_stage2)
        return {N_ADD: self.add, N_SUB: self.sub, N_MUL: self.mul}[k](a, b)

    def is_scalar(self, coords):
        return all(self.as_const(c) == 0 for c in coords[1:])

    def ext_mul(self, a, b, d, w, karatsuba):
        # src/graph.rs:448-507
        if self.is_scalar(a):
            return [self.mul(a[0], bk) for bk in b]
        if self.is_scalar(b):
            return [self.mul(b[0], ak) for ak in a]
        if d == 2 and karatsuba:
            p0 = self.mul(a[0], b[0])
            p1 = self.mul(a[1], b[1])
            sa = self.add(a[0], a[1])
            sb = self.add(b[0], b[1])
            s = self.mul(sa, sb)
            wn = self.constant(w)
            wp1 = self.mul(wn, p1)
            c0 = self.add(p0, wp1)
            t = self.sub(s, p0)
            c1 = self.sub(t, p1)
            return [c0, c1]
        wn = self.constant(w)
        out = []
        for k in range(d):
            low = high = None
            for i, ai in enumerate(a):
                for j, bj in enumerate(b):
                    if i + j == k:
                        term = self.mul(ai, bj)
                        low = term if low is None else self.add(low, term)
                    elif i + j == k + d:
                        term = self.mul(ai, bj)
                        high = term if high is None else self.add(high, term)
            out.append(low if high is None else self.add(low, self.mul(wn, high)))
        return out

    def expand_ext(self, e, spec, d, w, karatsuba):
        k = e.kind
        if k == "coords":
            if len(e.a) != d:
                raise CompileError("CoordsLength")
            return [self.compile_expr(c, spec, True) for c in e.a]
        if k == "base":
            zero = self.constant(0)
            coords = [zero] * d
            coords[0] = self.compile_expr(e.a, spec, True)
            return coords
        if k == "neg":
            return [self.neg(c) for c in self.expand_ext(e.a, spec, d, w, karatsuba)]
        a = self.expand_ext(e.a, spec, d, w, karatsuba)
        b = self.expand_ext(e.b, spec, d, w, karatsuba)
        if k == "add":
            return [self.add(a[i], b[i]) for i in range(d)]
        if k == "sub":
            return [self.sub(a[i], b[i]) for i in range(d)]
        return self.ext_mul(a, b, d, w, karatsuba)


class CompiledCircuit:
    """What `graph::compile` returns (src/graph.rs:62-76) plus the widths System::new keeps."""

    def __init__(self, nodes, zeros, lookups, main_width, preprocessed):
        self.nodes, self.zeros, self.lookups = nodes, zeros, lookups
        self.main_width = main_width
        self.preprocessed = preprocessed  # None or np.ndarray (h, w) uint64
        self.lookup_prefix_len = 0


class CircuitInputs:
    """src/system.rs:29-47."""

    def __init__(self, main_width=0, preprocessed=None, constraints=None, ext_constraints=None, lookups=None):
        self.main_width = main_width
        self.preprocessed = preprocessed
        self.constraints = list(constraints or [])
        self.ext_constraints = list(ext_constraints or [])
        self.lookups = list(lookups or [])


def compile_circuit(inputs, d=None, w=None):
    """`graph::compile` (src/graph.rs:120-188) on the spec System::new builds (src/system.rs:128-149)."""
    d = EXT_D if d is None else d
    w = EXT_W if w is None else w
    pre = inputs.preprocessed
    spec = {
        "main_width": inputs.main_width,
        "preprocessed_width": 0 if pre is None else int(pre.shape[1]),
        "stage2_width": max(len(inputs.lookups), 1) * d,
        "num_publics": 4 * d,
    }
    it = _Interner()
    lookups = []
    for lk in inputs.lookups:
        m = it.compile_expr(lk.multiplicity, spec, False)
        args = [it.compile_expr(a, spec, False) for a in lk.args]
        lookups.append((m, args))
    prefix = len(it.nodes)
    zeros = []

    def record(root, what):
        c = it.as_const(root)
        if c is None:
            zeros.append(root)
        elif c != 0:
            raise CompileError("UnsatisfiableConstant %s" % what)

    for i, c in enumerate(inputs.constraints):
        record(it.compile_expr(c, spec, False), "constraint %d" % i)
    for i, c in enumerate(inputs.ext_constraints):
        if c.is_purely_base():
            raise CompileError("PurelyBaseExtConstraint %d" % i)
        for k, root in enumerate(it.expand_ext(c, spec, d, w, d == 2)):
            record(root, "ext constraint %d coord %d" % (i, k))
    zeros = sorted(set(zeros))
    cc = CompiledCircuit(it.nodes, zeros, lookups, inputs.main_width, pre)
    cc.lookup_prefix_len = prefix
    return cc


class AirBuilder:
    """Recording builder, src/p3_adapter.rs:245-288 (+ p3-air default methods)."""

    def __init__(self, main_width, preprocessed_width=0, _cond=None, _sink=None):
        self.main_width, self.preprocessed_width = main_width, preprocessed_width
        self.constraints = [] if _sink is None else _sink
        self._cond = _cond

    def main(self):
        return [Expr.main(i) for i in range(self.main_width)], [Expr.main_next(i) for i in range(self.main_width)]

    def preprocessed(self):
        return ([Expr.var(SRC_PRE, 0, i) for i in range(self.preprocessed_width)],
                [Expr.var(SRC_PRE, 1, i) for i in range(self.preprocessed_width)])

    def is_first_row(self):
        return IS_FIRST_ROW

    def is_last_row(self):
        return IS_LAST_ROW

    def is_transition(self):
        return IS_TRANSITION

    def assert_zero(self, x):
        x = Expr.lift(x)
        self.constraints.append(x if self._cond is None else self._cond * x)

    def assert_eq(self, x, y):
        self.assert_zero(Expr.lift(x) - Expr.lift(y))

    def assert_one(self, x):
        self.assert_zero(Expr.lift(x) - Expr.const(1))

    def assert_bool(self, x):
        x = Expr.lift(x)
        self.assert_zero((Expr.const(1) - x) * x)  # p3 bool_check = andn(self, self)

    def assert_bools(self, xs):
        for x in xs:
            self.assert_bool(x)

    def when(self, cond):
        cond = Expr.lift(cond)
        c = cond if self._cond is None else self._cond * cond
        return AirBuilder(self.main_width, self.preprocessed_width, c, self.constraints)

    def when_transition(self):
        return self.when(IS_TRANSITION)

    def when_first_row(self):
        return self.when(IS_FIRST_ROW)

    def when_last_row(self):
        return self.when(IS_LAST_ROW)


def lookup_air(main_width, eval_fn, lookups, preprocessed=None):
    """`LookupAir::new(air, lookups).into()` (src/p3_adapter.rs:295-354)."""
    pw = 0 if preprocessed is None else int(preprocessed.shape[1])
    b = AirBuilder(main_width, pw)
    if eval_fn is not None:
        eval_fn(b)
    return CircuitInputs(main_width, preprocessed, b.constraints, [], lookups)


class Params:
    """CommitmentParameters + FriParameters (src/types.rs:171-197)."""

    def __init__(self, log_blowup=1, cap_height=0, log_final_poly_len=0, max_log_arity=1, num_queries=64,
                 commit_proof_of_work_bits=0, query_proof_of_work_bits=0):
        self.log_blowup, self.cap_height = log_blowup, cap_height
        self.log_final_poly_len, self.max_log_arity, self.num_queries = log_final_poly_len, max_log_arity, num_queries
        self.commit_proof_of_work_bits, self.query_proof_of_work_bits = commit_proof_of_work_bits, query_proof_of_work_bits

    def words(self):
        return [self.log_blowup, self.cap_height, self.log_final_poly_len, self.max_log_arity, self.num_queries,
                self.commit_proof_of_work_bits, self.query_proof_of_work_bits]


def bench_params():
    """benches/multi_stark.rs:244-258."""
    return Params(2, 0, 0, 1, 100, 10, 10)


def test_params():
    """The parameters every in-tree test/example uses (e.g. src/test_circuits/u32_add.rs:195-207)."""
    return Params(1, 0, 0, 1, 64, 0, 0)


def system_blob(params, compiled, poseidon2=None):
    """Serialise (params, compiled circuits) into the little-endian u64-word blob both libraries parse. Under
    `field(BABYBEAR)` the parameters are followed by the 141 Poseidon2 round constants (8 x 16 external, 13 internal)."""
    words = [BLOB_MAGIC] + params.words()
    if BLOB_MAGIC == BABYBEAR["BLOB_MAGIC"]:
        k = np.asarray(poseidon2, dtype=np.uint64).reshape(-1)
        if k.size != 141 or int(k.max()) >= P:
            raise ValueError("the BabyBear configuration needs 141 canonical Poseidon2 round constants")
        words += [int(x) for x in k]
    words += [len(compiled)]
    chunks = []
    for c in compiled:
        pre = c.preprocessed
        pw = 0 if pre is None else int(pre.shape[1])
        ph = 0 if pre is None else int(pre.shape[0])
        words += [c.main_width, pw, ph, len(c.nodes), len(c.zeros), len(c.lookups)]
        for (kind, source, offset, a, b) in c.nodes:
            words += [kind | (source << 8) | (offset << 16), a, b]
        words += list(c.zeros)
        for (m, args) in c.lookups:
            words += [m, len(args)] + list(args)
        chunks.append(np.asarray(words, dtype=np.uint64).tobytes())
        words = []
        if pre is not None:
            chunks.append(np.ascontiguousarray(pre, dtype=np.uint64).tobytes())
    chunks.append(np.asarray(words, dtype=np.uint64).tobytes())
    return b"".join(chunks)


# --------------------------------------------------------------------------- circuits
def u32_add_system_inputs():
    """[ByteTable, U32Add] -- benches/multi_stark.rs:73-165,260-267 (same as src/test_circuits/u32_add.rs)."""
    byte_index, u32_index = Expr.const(0), Expr.const(1)
    var = Expr.main
    byte_pre = np.arange(256, dtype=np.uint64).reshape(256, 1)
    byte_table = lookup_air(1, None, [Lookup.pull(var(0), [byte_index, Expr.preprocessed(0)])], byte_pre)

    def eval_add(b):
        local, _ = b.main()
        x, y, z, carry = local[0:4], local[4:8], local[8:12], local[12]
        b.assert_bool(carry)
        e1 = (x[0] + x[1] * Expr.const(256) + x[2] * Expr.const(256 ** 2) + x[3] * Expr.const(256 ** 3)
              + y[0] + y[1] * Expr.const(256) + y[2] * Expr.const(256 ** 2) + y[3] * Expr.const(256 ** 3))
        e2 = (z[0] + z[1] * Expr.const(256) + z[2] * Expr.const(256 ** 2) + z[3] * Expr.const(256 ** 3)
              + carry * Expr.const(256 ** 4))
        b.assert_eq(e1, e2)

    def word(i):
        return (var(i) + var(i + 1) * Expr.const(256) + var(i + 2) * Expr.const(256 ** 2)
                + var(i + 3) * Expr.const(256 ** 3))

    lookups = [Lookup.pull(var(13), [u32_index, word(0), word(4), word(8)])]
    lookups += [Lookup.push(Expr.const(1), [byte_index, var(i)]) for i in range(12)]
    u32_add = lookup_air(14, eval_add, lookups)
    return [byte_table, u32_add]


def u32_add_witness(pairs, height=None):
    """Traces for explicit (x, y) calls: src/test_circuits/u32_add.rs:150-190."""
    n = len(pairs)
    h = height or max(1, 1 << (n - 1).bit_length())
    x = np.array([p[0] for p in pairs], dtype=np.uint64)
    y = np.array([p[1] for p in pairs], dtype=np.uint64)
    return _u32_add_traces(x, y, h)


def _u32_add_traces(x, y, h):
    n = len(x)
    s = x + y
    z = s & np.uint64(0xFFFFFFFF)
    carry = s >> np.uint64(32)
    add = np.zeros((h, 14), dtype=np.uint64)
    for k in range(4):
        sh = np.uint64(8 * k)
        add[:n, k] = (x >> sh) & np.uint64(0xFF)
        add[:n, 4 + k] = (y >> sh) & np.uint64(0xFF)
        add[:n, 8 + k] = (z >> sh) & np.uint64(0xFF)
    add[:n, 12] = carry
    add[:n, 13] = 1
    counts = np.bincount(add[:n, :12].astype(np.int64).ravel(), minlength=256).astype(np.uint64)
    byte = counts.reshape(256, 1)
    claims = np.stack([np.ones(n, dtype=np.uint64), x, y, z], axis=1)
    return [byte, add], claims


def xorshift_pairs(num_adds, a0=0xDEADBEEF, b0=0xCAFEBABE):
    """benches/multi_stark.rs:180-192: two independent xorshift32 streams."""
    xs = np.empty(num_adds, dtype=np.uint64)
    ys = np.empty(num_adds, dtype=np.uint64)
    a, b, M = a0, b0, 0xFFFFFFFF
    for i in range(num_adds):
        a ^= (a << 13) & M
        a ^= a >> 17
        a ^= (a << 5) & M
        b ^= (b << 13) & M
        b ^= b >> 17
        b ^= (b << 5) & M
        xs[i] = a
        ys[i] = b
    return xs, ys


def u32_add_bench_witness(num_adds, a0=0xDEADBEEF, b0=0xCAFEBABE):
    """build_witness + build_claims (benches/multi_stark.rs:171-238): returns ([byte_trace, add_trace], claims)."""
    xs, ys = xorshift_pairs(num_adds, a0, b0)
    h = max(1, 1 << (num_adds - 1).bit_length())
    return _u32_add_traces(xs, ys, h)


def multi_u32_add_system_inputs(k):
    """[ByteTable, U32Add x k]: SURVEY §8d config 3 as ONE system (all AIRs in one proof)."""
    base = u32_add_system_inputs()
    return [base[0]] + [u32_add_system_inputs()[1] for _ in range(k)]


def multi_u32_add_witness(k, num_adds):
    """Per-AIR xorshift seeds a0 ^ (i * 0x9e3779b9), b0 ^ (i * 0x85ebca6b); byte multiplicities summed; all claims."""
    byte = np.zeros((256, 1), dtype=np.uint64)
    traces, claims = [None], []
    for i in range(k):
        a0 = 0xDEADBEEF ^ ((i * 0x9E3779B9) & 0xFFFFFFFF)
        b0 = 0xCAFEBABE ^ ((i * 0x85EBCA6B) & 0xFFFFFFFF)
        (bt, add), cl = u32_add_bench_witness(num_adds, a0, b0)
        byte += bt
        traces.append(add)
        claims.append(cl)
    traces[0] = byte
    return traces, np.concatenate(claims, axis=0)


def byte_operations_inputs():
    """ByteCS of src/test_circuits/byte_operations.rs:12-103: a 2^16-row preprocessed table [A, B, A^B, A&B, A|B], four
    multiplicity columns, no AIR constraints, four pull lookups (xor / and / or with 4 arguments, the pair range check
    with 3)."""
    a = np.repeat(np.arange(256, dtype=np.uint64), 256)
    b = np.tile(np.arange(256, dtype=np.uint64), 256)
    pre = np.stack([a, b, a ^ b, a & b, a | b], axis=1)
    var, pvar = Expr.main, Expr.preprocessed
    lookups = [Lookup.pull(var(i), [Expr.const(i), pvar(0), pvar(1), pvar(2 + i)]) for i in range(3)]
    lookups.append(Lookup.pull(var(3), [Expr.const(3), pvar(0), pvar(1)]))
    return [lookup_air(4, None, lookups, pre)]


def byte_operations_witness(calls):
    """ByteCalls::witness (src/test_circuits/byte_operations.rs:106-122): calls = [(op, x, y)], op 0 xor / 1 and / 2 or /
    3 pair range check. Returns ([trace], claims) with the claims of byte_test (:148-154), of ragged lengths."""
    trace = np.zeros((65536, 4), dtype=np.uint64)
    claims = []
    for (op, x, y) in calls:
        trace[256 * x + y, op] += 1
        claims.append([op, x, y] + ([x ^ y, x & y, x | y][op:op + 1] if op < 3 else []))
    return [trace], claims


def squares_inputs():
    """[RangeTable, Squares] of examples/preprocessed_proof.rs:27-88: a byte table pulled by its multiplicity column
    (one-argument lookups, no circuit index) and a squaring circuit [x, x^2, low, high, mult] pushing both bytes."""
    var = Expr.main
    pre = np.arange(256, dtype=np.uint64).reshape(256, 1)
    table = lookup_air(1, None, [Lookup.pull(var(0), [Expr.preprocessed(0)])], pre)

    def ev(b):
        local, _ = b.main()
        x, sq, low, high = local[0], local[1], local[2], local[3]
        b.assert_eq(sq, x * x)
        b.assert_eq(sq, low + high * Expr.const(256))

    squares = lookup_air(5, ev, [Lookup.push(var(4), [var(2)]), Lookup.push(var(4), [var(3)])])
    return [table, squares]


def squares_traces(n=16):
    """examples/preprocessed_proof.rs:108-127: squares of 0..n (n a power of two, n <= 256 so x^2 fits two bytes)."""
    x = np.arange(n, dtype=np.uint64)
    sq = x * x
    low, high = sq & np.uint64(0xFF), (sq >> np.uint64(8)) & np.uint64(0xFF)
    mult = np.bincount(np.concatenate([low, high]).astype(np.int64), minlength=256).astype(np.uint64).reshape(256, 1)
    return [mult, np.stack([x, sq, low, high, np.ones(n, dtype=np.uint64)], axis=1)]


def pythagorean_inputs():
    """examples/simple_proof.rs:21-44."""

    def ev(b):
        local, _ = b.main()
        b.assert_eq(local[0] * local[0] + local[1] * local[1], local[2] * local[2])

    return [lookup_air(3, ev, [])]


def pythagorean_trace(rows):
    """examples/simple_proof.rs:64-83, the 4 triples cycled to `rows` rows."""
    base = np.array([[3, 4, 5], [5, 12, 13], [8, 15, 17], [7, 24, 25]], dtype=np.uint64)
    reps = (rows + 3) // 4
    return np.tile(base, (reps, 1))[:rows].copy()


def verifier_test_inputs():
    """[CS::Pythagorean, CS::Complex] of the verifier's own tests, src/verifier.rs:718-782: a (a^2 + b^2) = a c^2 (the extra
    factor raises the constraint degree to 3) and the complex product (a + ib)(c + id) = e + if; no lookups."""
    def pyth(b):
        local, _ = b.main()
        b.assert_eq(local[0] * (local[0] * local[0] + local[1] * local[1]), local[0] * (local[2] * local[2]))

    def cplx(b):
        local, _ = b.main()
        b.assert_eq(local[0] * local[2] - local[1] * local[3], local[4])
        b.assert_eq(local[0] * local[3] + local[1] * local[2], local[5])

    return [lookup_air(3, pyth, []), lookup_air(6, cplx, [])]


def verifier_test_traces(doublings=0):
    """src/verifier.rs:787-797 (4 + 2 rows); `doublings` = 4 gives the 16-row traces of :806-813 (one row repeated)"""
    if doublings:
        py, cx = [3, 4, 5], [4, 2, 3, 1, 10, 10]
        for _ in range(doublings):
            py, cx = py + py, cx + cx
        return [np.array(py, dtype=np.uint64).reshape(-1, 3), np.array(cx, dtype=np.uint64).reshape(-1, 6)]
    return [np.array([3, 4, 5, 5, 12, 13, 8, 15, 17, 7, 24, 25], dtype=np.uint64).reshape(4, 3),
            np.array([4, 2, 3, 1, 10, 10, 3, 2, 5, 1, 13, 13], dtype=np.uint64).reshape(2, 6)]


def even_odd_inputs(with_dead=False):
    """src/lookup.rs:868-947 (Even / Odd / Dead circuits)."""
    var = Expr.main

    def ev(b):
        local, _ = b.main()
        mult, inp, inp_inv, is_zero, not_zero = local[0], local[1], local[2], local[3], local[4]
        b.assert_bools([is_zero, not_zero])
        b.when(mult).assert_one(is_zero + not_zero)
        b.when(is_zero).assert_zero(inp)
        b.when(not_zero).assert_one(inp * inp_inv)

    mult, inp, is_zero, not_zero, rec = var(0), var(1), var(3), var(4), var(5)
    even_i, odd_i, one = Expr.const(0), Expr.const(1), Expr.const(1)
    even = [Lookup.pull(mult, [even_i, inp, not_zero * rec + is_zero]), Lookup.push(not_zero, [odd_i, inp - one, rec])]
    odd = [Lookup.pull(mult, [odd_i, inp, not_zero * rec]), Lookup.push(not_zero, [even_i, inp - one, rec])]
    out = [lookup_air(6, ev, even), lookup_air(6, ev, odd)]
    if with_dead:
        out.append(lookup_air(6, ev, [Lookup.pull(mult, [Expr.const(2), inp])]))
    return out


def even_odd_traces():
    """src/lookup.rs:975-1007; claim is [0, 4, 1]."""
    inv = lambda v: pow(v, P - 2, P)  # noqa: E731
    even = np.array([[1, 4, inv(4), 0, 1, 1], [1, 2, inv(2), 0, 1, 1], [1, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0]], dtype=np.uint64)
    odd = np.array([[1, 3, inv(3), 0, 1, 1], [1, 1, inv(1), 0, 1, 1], [0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0]], dtype=np.uint64)
    return [even, odd]


# --------------------------------------------------------------------------- BabyBear / Poseidon2 configuration
def poseidon2_constants_stand_in(seed=42):
    """141 uniform values below p from numpy's PCG64: the stand-in constants of rounds 1-3 (kept for the fuzzers, which want many
    different permutations, and as the second instantiation the parity tests run under)."""
    return np.random.default_rng(seed).integers(0, BABYBEAR["P"], 141, dtype=np.uint64)


class SmallRngXoshiro:
    """rand's `SmallRng` on a 64-bit target = Xoshiro256++, seeded by `seed_from_u64` through SplitMix64, `next_u32` = the HIGH half
    of `next_u64` [UPSTREAM-RECALL: rand 0.9 `rngs::xoshiro256plusplus`; the reference pins rand 0.10.2 (Cargo.lock:864-865), whose
    SmallRng is assumed unchanged - the pinning kit's 141 dumped constants settle it, tests/test_reference_pins.py]. The
    generator itself is the published xoshiro256++ 1.0 (Blackman / Vigna), checked against its reference vector in
    tests/test_frontend.py."""

    MASK = (1 << 64) - 1

    def __init__(self, seed_u64):
        s, st = [], seed_u64 & self.MASK
        for _ in range(4):  # SplitMix64: one output per 8 seed bytes, little-endian words = the state words in order
            st = (st + 0x9E3779B97F4A7C15) & self.MASK
            z = st
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.MASK
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.MASK
            s.append(z ^ (z >> 31))
        self.s = s

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & SmallRngXoshiro.MASK

    def next_u64(self):
        s = self.s
        result = (self._rotl((s[0] + s[3]) & self.MASK, 23) + s[0]) & self.MASK
        t = (s[1] << 17) & self.MASK
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 45)
        return result

    def next_u32(self):
        return self.next_u64() >> 32


def poseidon2_constants_small_rng(seed=42):
    """The reference's own instantiation, restated: `Perm::new_from_rng_128(&mut SmallRng::seed_from_u64(42))`
    (src/test_circuits/baby_bear_config.rs:54-55) [UPSTREAM-RECALL for everything below; nothing in the reference's tree pins it]:
      * p3-poseidon2 `Poseidon2::new_from_rng_128` -> round numbers for BabyBear, width 16, S-box degree 7: 8 full rounds, 13
        partial ones; `ExternalLayerConstants::new_from_rng` draws the 4 initial rounds' [F; 16] arrays, then the 4 terminal
        ones, element by element, and the 13 internal constants follow;
      * p3-monty-31 `Distribution<MontyField31> for StandardUniform`: `loop { let x = rng.next_u32() >> 1; if x < P { return
        MontyField31::new_monty(x) } }` - the 31-bit value is taken AS the Montgomery word, so the field element is x * 2^-32.
    Returned as canonical values in the front-end's order (initial, terminal, internal), which is what the ABI takes."""
    p = BABYBEAR["P"]
    rng = SmallRngXoshiro(seed)
    r_inv = pow(1 << 32, -1, p)
    out = []
    while len(out) < 141:
        x = rng.next_u32() >> 1
        if x < p:
            out.append(x * r_inv % p)
    return np.array(out, dtype=np.uint64)


def poseidon2_constants(seed=42):
    """141 round constants for Poseidon2BabyBear<16> (8 x 16 external: initial then terminal; then 13 internal), canonical values.
    Default = the restatement of the reference's instantiation (poseidon2_constants_small_rng: SmallRng::seed_from_u64(42) ->
    new_from_rng_128, labelled UPSTREAM-RECALL and checked by the pinning kit's dumped constants when they arrive); they remain
    INPUTS of the library (msbb_system_create takes them), so any other instantiation works the same."""
    return poseidon2_constants_small_rng(seed)


def mul_air_inputs():
    """MulAir + its self-cancelling push/pull pair, src/test_circuits/baby_bear_config.rs:129-157: a * b = c per row,
    lookups push(1, [a, c]) and pull(1, [a, c]). Use under `field(BABYBEAR)`."""
    def ev(b):
        local, _ = b.main()
        b.assert_eq(local[0] * local[1], local[2])

    one = Expr.const(1)
    return [lookup_air(3, ev, [Lookup.push(one, [Expr.main(0), Expr.main(2)]), Lookup.pull(one, [Expr.main(0), Expr.main(2)])])]


def mul_air_trace(rows):
    """rows (r + 1, r + 2, product mod p): BASELINE config 4's scaling of the reference's 4-row trace
    (baby_bear_config.rs:176-192 uses (2,3,6), (4,5,20), (7,8,56), (0,0,0))."""
    r = np.arange(rows, dtype=np.uint64)
    a, b = (r + 1) % P, (r + 2) % P
    return np.stack([a, b, (a * b) % P], axis=1).astype(np.uint64)


def mul_air_smoke_trace():
    """the reference's own 4 rows, baby_bear_config.rs:176-192"""
    return np.array([[2, 3, 6], [4, 5, 20], [7, 8, 56], [0, 0, 0]], dtype=np.uint64)


def pack_claims(claims):
    """list of sequences (or 2-D array) -> (offsets uint64[n+1], data uint64[total])."""
    if isinstance(claims, np.ndarray) and claims.ndim == 2:
        n, w = claims.shape
        return np.arange(0, (n + 1) * w, w, dtype=np.uint64), np.ascontiguousarray(claims.astype(np.uint64) % np.uint64(P)).ravel()
    offs = np.zeros(len(claims) + 1, dtype=np.uint64)
    for i, c in enumerate(claims):
        offs[i + 1] = offs[i] + len(c)
    data = np.array([int(x) % P for c in claims for x in c], dtype=np.uint64)
    return offs, data
