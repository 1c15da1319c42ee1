"""The reference's largest scenario, authored in this front-end: the BLAKE3 compression system of
/root/reference/src/test_circuits/blake3.rs - nine circuits linked by lookups (a 2^16-row byte-pair table with a preprocessed
trace, u32 xor / add, the four rotations, the G function, and the 2625-column compression circuit), their witness generation
(`Blake3CompressionClaims::witness`, :1511-2213) and the hasher that records every compression of a BLAKE3 hash (:32-352).

Authoring-time code, like the rest of frontend.py: it exists to FEED the prover path the reference's own heaviest test
(`test_compression_reference_compatibility`, `test_all_claims`, :2215-2613). Lookup channels are numbered by
`Blake3CompressionCircuit::position` (:426-441): the byte-pair table serves channel 0 (xor) and channel 7 (pair range check)."""
import numpy as np

from .frontend import Expr, Lookup, lookup_air

IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
MSG_PERMUTATION = [2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8]
# the eight G applications of a round on the 32-word state [cv 8 | iv 4 | counter 2, block_len, flags | message 16] (:360-365)
G_A, G_B, G_C, G_D = [0, 1, 2, 3, 0, 1, 2, 3], [4, 5, 6, 7, 5, 6, 7, 4], [8, 9, 10, 11, 10, 11, 8, 9], [12, 13, 14, 15, 15, 12, 13, 14]
G_MX, G_MY = [16, 18, 20, 22, 24, 26, 28, 30], [17, 19, 21, 23, 25, 27, 29, 31]
CHUNK_START, CHUNK_END, PARENT, ROOT = 1, 2, 4, 8
M32 = 0xFFFFFFFF

# lookup channels = Blake3CompressionCircuit::position (:426-441)
U8_XOR, U32_XOR, U32_ADD, ROT8, ROT16, ROT12, ROT7, U8_PAIR, G_FUNCTION, COMPRESSION = range(10)
WIDTHS = {"u8": 2, "u32_xor": 13, "u32_add": 14, "rot8": 9, "rot16": 9, "rot12": 25, "rot7": 25, "g": 81, "compression": 2625}  # :369-411


def _rotr(x, k):
    return ((x >> k) | (x << (32 - k))) & M32


def _g(a, b, c, d, mx, my):
    """one G application; returns every intermediate the G-function circuit keeps a column group for"""
    a0t = (a + b) & M32
    a0 = (a0t + mx) & M32
    d0t = d ^ a0
    d0 = _rotr(d0t, 16)
    c0 = (c + d0) & M32
    b0t = b ^ c0
    b0 = _rotr(b0t, 12)
    a1t = (a0 + b0) & M32
    a1 = (a1t + my) & M32
    d1t = d0 ^ a1
    d1 = _rotr(d1t, 8)
    c1 = (c0 + d1) & M32
    b1t = b0 ^ c1
    b1 = _rotr(b1t, 7)
    return a0t, a0, d0t, d0, c0, b0t, b0, a1t, a1, d1t, d1, c1, b1t, b1


def _rounds(state, visit=None):
    """the seven rounds on the 32-word state (in place); visit(inputs 6, outputs a1 b1 c1 d1) per G application"""
    for r in range(7):
        for j in range(8):
            ins = (state[G_A[j]], state[G_B[j]], state[G_C[j]], state[G_D[j]], state[G_MX[j]], state[G_MY[j]])
            t = _g(*ins)
            a1, d1, c1, b1 = t[8], t[10], t[11], t[13]
            state[G_A[j]], state[G_B[j]], state[G_C[j]], state[G_D[j]] = a1, b1, c1, d1
            if visit:
                visit(ins, (a1, b1, c1, d1))
        if r < 6:
            state[16:32] = [state[16 + MSG_PERMUTATION[i]] for i in range(16)]


def compress(cv, block_words, counter, block_len, flags):
    """the BLAKE3 compression function: 16 output words (:47-116)"""
    state = list(cv) + IV[:4] + [counter & M32, (counter >> 32) & M32, block_len, flags] + list(block_words)
    _rounds(state)
    for i in range(8):
        state[i] ^= state[i + 8]
        state[i + 8] ^= cv[i]
    return state[:16]


def blake3_compressions(data: bytes):
    """(every compression of BLAKE3(data) in the order the reference implementation performs them, the 32-byte digest) -
    `blake3_new_update_finalize` (:32-352). A compression is a dict cv / block_words / counter_low / counter_high / block_len /
    flags / output."""
    infos = []

    def run(cv, words, counter, block_len, flags):
        out = compress(cv, words, counter, block_len, flags)
        infos.append({"cv": list(cv), "block_words": list(words), "counter_low": counter & M32, "counter_high": (counter >> 32) & M32,
                      "block_len": block_len, "flags": flags, "output": out})
        return out

    def words_of(block):
        b = block + bytes(64 - len(block))
        return [int.from_bytes(b[4 * i:4 * i + 4], "little") for i in range(16)]

    chunks = [data[i:i + 1024] for i in range(0, len(data), 1024)] or [b""]
    stack, node = [], None
    for ci, chunk in enumerate(chunks):
        blocks = [chunk[i:i + 64] for i in range(0, len(chunk), 64)] or [b""]
        cv = IV
        for bi, blk in enumerate(blocks[:-1]):
            cv = run(cv, words_of(blk), ci, 64, CHUNK_START if bi == 0 else 0)[:8]
        last = (cv, words_of(blocks[-1]), ci, len(blocks[-1]), (CHUNK_START if len(blocks) == 1 else 0) | CHUNK_END)
        if ci + 1 < len(chunks):  # a complete chunk with input behind it: its chaining value joins the stack of sub-tree roots
            new_cv, total = run(*last)[:8], ci + 1
            while total & 1 == 0:
                new_cv = run(IV, stack.pop() + new_cv, 0, 64, PARENT)[:8]
                total >>= 1
            stack.append(new_cv)
        else:
            node = last
    while stack:
        right = run(*node)[:8]
        node = (IV, stack.pop() + right, 0, 64, PARENT)
    out = run(node[0], node[1], 0, node[3], node[4] | ROOT)
    return infos, b"".join(w.to_bytes(4, "little") for w in out[:8])


def compression_claim(info):
    """[channel 9, state_in (32 words: cv, IV[..4], counter, block_len, flags, message), state_out (16 words)] (:2222-2239, :2319-2329)"""
    return [COMPRESSION] + info["cv"] + IV[:4] + [info["counter_low"], info["counter_high"], info["block_len"], info["flags"]] + \
        info["block_words"] + info["output"]


# ---------------------------------------------------------------------------------------------------------------------
def _c(v):
    return Expr.const(v)


def _word(var, i0, i1, i2, i3):
    return var(i0) + var(i1) * _c(256) + var(i2) * _c(256 * 256) + var(i3) * _c(256 * 256 * 256)


def _w(i):  # the u32 held little-endian in main columns i .. i + 3
    return _word(Expr.main, i, i + 1, i + 2, i + 3)


def _eval_u32_add(b):  # :501-526
    local, _ = b.main()
    x, y, z, carry = local[0:4], local[4:8], local[8:12], local[12]
    b.assert_bool(carry)
    e1 = (x[0] + x[1] * _c(256) + x[2] * _c(256 * 256) + x[3] * _c(256 * 256 * 256)
          + y[0] + y[1] * _c(256) + y[2] * _c(256 * 256) + y[3] * _c(256 * 256 * 256))
    e2 = z[0] + z[1] * _c(256) + z[2] * _c(256 * 256) + z[3] * _c(256 * 256 * 256) + carry * _c(256 ** 4)
    b.assert_eq(e1, e2)


def _eval_rot_7_12(b):  # :527-565: input = div 2^k + rem, output = div + rem 2^(32-k) (the FIXME there: not range checked)
    value, output, two_k, two_32k, div, rem = (_w(i) for i in (1, 5, 9, 13, 17, 21))
    b.assert_eq(value, div * two_k + rem)
    b.assert_eq(output, div + rem * two_32k)


def _eval_compression(b):  # :566-770
    state = [_w(1 + 4 * i) for i in range(32)]
    cv_expected = state[0:8]
    off = 129
    g_cols = []
    for _ in range(56):  # a_in b_in c_in d_in mx_in my_in | a_1 d_1 c_1 b_1
        g_cols.append([_w(off + 4 * k) for k in range(10)])
        off += 40
    xor_cols = []
    for _ in range(8):  # state_i, state_i_8, their xor | state_i_8 again, chaining value, their xor
        xor_cols.append([_w(off + 4 * k) for k in range(6)])
        off += 24
    state_out = [_w(off + 4 * k) for k in range(16)]
    k = 0
    for r in range(7):
        for j in range(8):
            a_in, b_in, c_in, d_in, mx_in, my_in, a_1, d_1, c_1, b_1 = g_cols[k]
            for want, got in ((state[G_A[j]], a_in), (state[G_B[j]], b_in), (state[G_C[j]], c_in), (state[G_D[j]], d_in),
                              (state[G_MX[j]], mx_in), (state[G_MY[j]], my_in)):
                b.assert_eq(want, got)
            state[G_A[j]], state[G_B[j]], state[G_C[j]], state[G_D[j]] = a_1, b_1, c_1, d_1
            k += 1
        if r < 6:
            state[16:32] = [state[16 + MSG_PERMUTATION[i]] for i in range(16)]
    for i in range(8):
        s_i, s_i8, x_i, s_i8_again, cv_i, x_cv = xor_cols[i]
        b.assert_eq(state[i], s_i)
        b.assert_eq(state[i + 8], s_i8)
        b.assert_eq(x_i, state_out[i])
        b.assert_eq(state[i + 8], s_i8_again)
        b.assert_eq(cv_expected[i], cv_i)
        b.assert_eq(x_cv, state_out[i + 8])


def blake3_system_inputs():
    """The nine LookupAirs in the order `System::new` receives them (:2258-2312)."""
    var, pvar, one, zero = Expr.main, Expr.preprocessed, _c(1), _c(0)
    a = np.repeat(np.arange(256, dtype=np.uint64), 256)
    bb = np.tile(np.arange(256, dtype=np.uint64), 256)
    pre = np.stack([a, bb, a ^ bb], axis=1)  # :458-474
    u8 = lookup_air(2, None, [Lookup.pull(var(0), [_c(U8_XOR), pvar(0), pvar(1), pvar(2)]),
                              Lookup.pull(var(1), [_c(U8_PAIR), pvar(0), pvar(1)])], pre)

    def pair(i, j):
        return Lookup.push(one, [_c(U8_PAIR), var(i), var(j)])

    u32_xor = lookup_air(13, None, [Lookup.pull(var(0), [_c(U32_XOR), _w(1), _w(5), _w(9)])] +
                         [Lookup.push(one, [_c(U8_XOR), var(i + 1), var(i + 5), var(i + 9)]) for i in range(4)])
    u32_add = lookup_air(14, _eval_u32_add, [Lookup.pull(var(13), [_c(U32_ADD), _w(0), _w(4), _w(8)])] +
                         [pair(i, i + 4) for i in range(4)] + [Lookup.push(one, [_c(U8_PAIR), var(i + 8), zero]) for i in range(4)])
    rot8 = lookup_air(9, None, [Lookup.pull(var(0), [_c(ROT8), _w(1), _word(var, 2, 3, 4, 1)])] + [pair(i + 1, i + 3) for i in range(2)])
    rot16 = lookup_air(9, None, [Lookup.pull(var(0), [_c(ROT16), _w(1), _word(var, 3, 4, 1, 2)])] + [pair(i + 1, i + 3) for i in range(2)])
    rot12 = lookup_air(25, _eval_rot_7_12, [Lookup.pull(var(0), [_c(ROT12), _w(1), _w(5)])])
    rot7 = lookup_air(25, _eval_rot_7_12, [Lookup.pull(var(0), [_c(ROT7), _w(1), _w(5)])])
    # G: multiplicity, a_in b_in c_in d_in mx_in my_in (1..24), a_0_tmp 25, a_0 29, d_0_tmp 33, d_0 37, c_0 41, b_0_tmp 45, b_0 49,
    # a_1_tmp 53, a_1 57, d_1_tmp 61, d_1 65, c_1 69, b_1_tmp 73, b_1 77 (:1107-1358)
    steps = [(U32_ADD, 1, 5, 25), (U32_ADD, 25, 17, 29), (U32_XOR, 13, 29, 33), (ROT16, 33, 37), (U32_ADD, 9, 37, 41), (U32_XOR, 5, 41, 45),
             (ROT12, 45, 49), (U32_ADD, 29, 49, 53), (U32_ADD, 53, 21, 57), (U32_XOR, 37, 57, 61), (ROT8, 61, 65), (U32_ADD, 41, 65, 69),
             (U32_XOR, 49, 69, 73), (ROT7, 73, 77)]
    g = lookup_air(81, None, [Lookup.pull(var(0), [_c(G_FUNCTION)] + [_w(i) for i in (1, 5, 9, 13, 17, 21, 57, 65, 69, 77)])] +
                   [Lookup.push(one, [_c(s[0])] + [_w(i) for i in s[1:]]) for s in steps])
    comp = lookup_air(2625, _eval_compression,
                      [Lookup.pull(var(0), [_c(COMPRESSION)] + [_w(1 + 4 * i) for i in range(32)] + [_w(2561 + 4 * i) for i in range(16)])] +
                      [Lookup.push(one, [_c(G_FUNCTION)] + [_w(129 + 40 * k + 4 * i) for i in range(10)]) for k in range(56)] +
                      [Lookup.push(one, [_c(U32_XOR)] + [_w(2369 + 12 * k + 4 * i) for i in range(3)]) for k in range(16)])
    return [u8, u32_xor, u32_add, rot8, rot16, rot12, rot7, g, comp]


# ---------------------------------------------------------------------------------------------------------------------
def _bytes(*words):
    out = []
    for w in words:
        out += [w & 255, (w >> 8) & 255, (w >> 16) & 255, w >> 24]
    return out


def _matrix(rows, width):
    """rows (lists of ints) padded with zero rows to a power-of-two height; (matrix, number of padding rows). No rows at all
    gives the single zero row the reference starts from."""
    n = max(len(rows), 1)
    h = 1 << (n - 1).bit_length()
    m = np.zeros((h, width), dtype=np.uint64)
    if rows:
        m[:len(rows)] = np.asarray(rows, dtype=np.uint64)
    return m, h - len(rows)


def blake3_witness(claims):
    """`Blake3CompressionClaims::witness` (:1516-2213): the nine stage-1 traces for these claims. A claim is
    [channel, values..] with the channel of `Blake3CompressionCircuit::position`; what a circuit sends down to the circuits
    below it is appended to their work lists, padding rows included (their lookups are pushed with multiplicity one on every
    row, so a padding row sends all-zero tuples that the lower circuit has to absorb)."""
    work = {k: [] for k in range(10)}
    for c in claims:
        c = [int(x) for x in c]
        want = {U8_XOR: 4, U32_XOR: 4, U32_ADD: 4, ROT8: 3, ROT16: 3, ROT12: 3, ROT7: 3, U8_PAIR: 3, G_FUNCTION: 11, COMPRESSION: 49}.get(c[0] if c else -1)
        if want is None or len(c) != want:
            raise ValueError("wrong claim format")
        if c[0] == G_FUNCTION:  # the claim lists a_1, d_1, c_1, b_1 (:1608-1624)
            work[G_FUNCTION].append(tuple(c[1:7]) + (c[7], c[10], c[9], c[8]))
        elif c[0] == COMPRESSION:
            work[COMPRESSION].append((c[1:33], c[33:49]))
        else:
            work[c[0]].append(tuple(c[1:]))
    zero_g = (0,) * 10

    # ---- compression (:1650-1804)
    rows = []
    for state_in, state_out in work[COMPRESSION]:
        row = [1] + _bytes(*state_in)
        state = list(state_in)

        def visit(ins, outs, row=row):
            a1, b1, c1, d1 = outs
            work[G_FUNCTION].append(tuple(ins) + (a1, b1, c1, d1))
            row.extend(_bytes(*ins, a1, d1, c1, b1))

        _rounds(state, visit)
        for i in range(8):
            left, right = state[i], state[i + 8]
            state[i] ^= state[i + 8]
            row += _bytes(left, right, state[i])
            work[U32_XOR].append((left, right, state[i]))
            left, right = state[i + 8], state_in[i]
            state[i + 8] ^= state_in[i]
            row += _bytes(left, right, state[i + 8])
            work[U32_XOR].append((left, right, state[i + 8]))
        if state[:16] != list(state_out):
            raise ValueError("compression claim: state_out is not the compression of state_in")
        rows.append(row + _bytes(*state_out))
    comp, pad = _matrix(rows, WIDTHS["compression"])
    for _ in range(pad):
        work[G_FUNCTION] += [zero_g] * 56
        work[U32_XOR] += [(0, 0, 0)] * 16

    # ---- G function (:1806-1904)
    rows = []
    for (a, b, c, d, mx, my, a1, b1, c1, d1) in work[G_FUNCTION]:
        a0t, a0, d0t, d0, c0, b0t, b0, a1t, a_1, d1t, d_1, c_1, b1t, b_1 = _g(a, b, c, d, mx, my)
        if (a_1, b_1, c_1, d_1) != (a1, b1, c1, d1):
            raise ValueError("G-function claim: outputs do not match")
        work[U32_ADD] += [(a, b, a0t), (a0t, mx, a0)]
        work[U32_XOR].append((d, a0, d0t))
        work[ROT16].append((d0t, d0))
        work[U32_ADD].append((c, d0, c0))
        work[U32_XOR].append((b, c0, b0t))
        work[ROT12].append((b0t, b0))
        work[U32_ADD] += [(a0, b0, a1t), (a1t, my, a_1)]
        work[U32_XOR].append((d0, a_1, d1t))
        work[ROT8].append((d1t, d_1))
        work[U32_ADD].append((c0, d_1, c_1))
        work[U32_XOR].append((b0, c_1, b1t))
        work[ROT7].append((b1t, b_1))
        rows.append([1] + _bytes(a, b, c, d, mx, my, a0t, a0, d0t, d0, c0, b0t, b0, a1t, a_1, d1t, d_1, c_1, b1t, b_1))
    g, pad = _matrix(rows, WIDTHS["g"])
    for _ in range(pad):
        for ch in (ROT7, ROT8, ROT16, ROT12):
            work[ch].append((0, 0))
        work[U32_XOR] += [(0, 0, 0)] * 4
        work[U32_ADD] += [(0, 0, 0)] * 6

    # ---- u32 xor (:1906-1946): four byte triples per row go down to the byte-pair table's xor channel
    rows = []
    for (l, r, x) in work[U32_XOR]:
        if l ^ r != x:
            raise ValueError("u32 xor claim does not hold")
        lb, rb, xb = _bytes(l), _bytes(r), _bytes(x)
        rows.append([1] + lb + rb + xb)
        work[U8_XOR] += list(zip(lb, rb, xb))
    u32_xor, pad = _matrix(rows, WIDTHS["u32_xor"])
    work[U8_XOR] += [(0, 0, 0)] * (4 * pad)

    # ---- u32 add (:1948-1994): (x_i, y_i) and (z_i, 0) per byte go to the pair range check
    rows = []
    for (l, r, s) in work[U32_ADD]:
        if (l + r) & M32 != s:
            raise ValueError("u32 add claim does not hold")
        lb, rb, sb = _bytes(l), _bytes(r), _bytes(s)
        rows.append(lb + rb + sb + [(l + r) >> 32, 1])
        for i in range(4):
            work[U8_PAIR] += [(lb[i], rb[i]), (sb[i], 0)]
    u32_add, pad = _matrix(rows, WIDTHS["u32_add"])
    work[U8_PAIR] += [(0, 0)] * (8 * pad)

    # ---- rotations by whole bytes (:1996-2080): the input's bytes are range checked in pairs (0, 2), (1, 3)
    def rot_bytes(channel, k, name):
        rows = []
        for (v, rot) in work[channel]:
            if _rotr(v, k) != rot:
                raise ValueError("rotation claim does not hold")
            vb = _bytes(v)
            rows.append([1] + vb + _bytes(rot))
            work[U8_PAIR] += [(vb[0], vb[2]), (vb[1], vb[3])]
        m, pad = _matrix(rows, WIDTHS[name])
        work[U8_PAIR] += [(0, 0)] * (2 * pad)
        return m

    rot8, rot16 = rot_bytes(ROT8, 8, "rot8"), rot_bytes(ROT16, 16, "rot16")

    # ---- rotations by 12 and 7 bits (:2082-2140): value = div 2^k + rem, rotated = div + rem 2^(32 - k)
    def rot_bits(channel, k, name):
        rows = []
        for (v, rot) in work[channel]:
            if _rotr(v, k) != rot:
                raise ValueError("rotation claim does not hold")
            rows.append([1] + _bytes(v, rot, 1 << k, 1 << (32 - k), v >> k, v & ((1 << k) - 1)))
        return _matrix(rows, WIDTHS[name])[0]

    rot12, rot7 = rot_bits(ROT12, 12, "rot12"), rot_bits(ROT7, 7, "rot7")

    # ---- the byte-pair table (:2142-2176): multiplicities of (i, j, i ^ j) and of (i, j)
    u8 = np.zeros((65536, 2), dtype=np.uint64)
    for (i, j, x) in work[U8_XOR]:
        if 0 <= i < 256 and 0 <= j < 256 and x == (i ^ j):
            u8[256 * i + j, 0] += 1
    for (i, j) in work[U8_PAIR]:
        if 0 <= i < 256 and 0 <= j < 256:
            u8[256 * i + j, 1] += 1
    return [u8, u32_xor, u32_add, rot8, rot16, rot12, rot7, g, comp]


def all_claims_cases():
    """the claim sets of `test_all_claims` (:2343-2613), one system run each"""
    a8, b8 = 0xA1, 0xA8
    a, b = 0x000000FF, 0x0000FF01
    gi = (0x11111111, 0x22222222, 0x33333333, 0x44444444, 0x55555555, 0x66666666)
    t = _g(*gi)
    state_in = [0x1111 * i for i in range(16)] + [(0x1111 * i) << 16 for i in range(16)]
    state_out = [0xD304E51C, 0xC2DF34A0, 0x5EBA7F1F, 0x2AB9650F, 0xD9CEF159, 0x4E9D3A6A, 0xCAC2E310, 0xC6B9BE7E, 0xAD9FD58A, 0x0899E71B,
                 0xCA51A599, 0xC3FBD7C0, 0x751D2F26, 0x6CD0AC6B, 0xC58F3C1D, 0xE6D65414]
    return [
        ("u8_xor", [[U8_XOR, a8, b8, a8 ^ b8]]),
        ("u32_xor", [[U32_XOR, a, b, a ^ b]]),
        ("u32_add", [[U32_ADD, a, b, (a + b) & M32]]),
        ("rotations", [[ROT8, a, _rotr(a, 8)], [ROT16, a, _rotr(a, 16)], [ROT12, a, _rotr(a, 12)], [ROT7, a, _rotr(a, 7)]]),
        ("g_function", [[G_FUNCTION] + list(gi) + [t[8], t[10], t[11], t[13]]]),
        ("compression", [[COMPRESSION] + state_in + state_out]),
    ]
