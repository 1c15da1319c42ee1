"""CPU tests (no GPU): the oracle's arithmetic and hashing against the reference's own known-answer vectors and
against independent Python big-integer arithmetic."""
import hashlib
import ctypes as C

import numpy as np
import pytest

from conftest import P, rand_field


def test_goldilocks_constants(oracle):
    L = oracle.lib()
    # GENERATOR = 7 generates the multiplicative group: 7^((p-1)/q) != 1 for every prime q | p - 1 = 2^32 * 3 * 5 * 17 * 257 * 65537
    for q in (2, 3, 5, 17, 257, 65537):
        assert pow(7, (P - 1) // q, P) != 1
    g32 = L.mso_gl_two_adic_generator(32)
    assert g32 == pow(7, (P - 1) >> 32, P) == 1753635133440165772
    for bits in range(0, 33):
        g = L.mso_gl_two_adic_generator(bits)
        assert pow(g, 1 << bits, P) == 1 and (bits == 0 or pow(g, 1 << (bits - 1), P) == P - 1)


def test_goldilocks_arithmetic_matches_bigint(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    a, b = rand_field(rng, 3000), rand_field(rng, 3000)
    for x, y in zip(a.tolist(), b.tolist()):
        assert L.mso_gl_mul(x, y) == x * y % P
        assert L.mso_gl_add(x, y) == (x + y) % P
        assert L.mso_gl_sub(x, y) == (x - y) % P
    for x in a[:200].tolist():
        if x:
            assert L.mso_gl_mul(L.mso_gl_inv(x), x) == 1


def test_ext2_is_x2_minus_7(oracle):
    L = oracle.lib()
    u64p = C.POINTER(C.c_uint64)
    rng = np.random.default_rng(1)
    for _ in range(300):
        a, b = rand_field(rng, 2), rand_field(rng, 2)
        o = np.zeros(2, dtype=np.uint64)
        L.mso_e2_mul(a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), o.ctypes.data_as(u64p))
        a0, a1, b0, b1 = int(a[0]), int(a[1]), int(b[0]), int(b[1])
        assert (int(o[0]), int(o[1])) == ((a0 * b0 + 7 * a1 * b1) % P, (a0 * b1 + a1 * b0) % P)
        if a0 or a1:
            inv = np.zeros(2, dtype=np.uint64)
            L.mso_e2_inv(a.ctypes.data_as(u64p), inv.ctypes.data_as(u64p))
            L.mso_e2_mul(a.ctypes.data_as(u64p), inv.ctypes.data_as(u64p), o.ctypes.data_as(u64p))
            assert (int(o[0]), int(o[1])) == (1, 0)


# ---- BLAKE3: the reference's literal KATs
def test_blake3_g_function_kat(oracle):
    """/root/reference/src/test_circuits/blake3.rs:2616-2644"""
    v = np.array([0x11111111, 0x22222222, 0x33333333, 0x44444444], dtype=np.uint32)
    oracle.lib().mso_b3_g(v.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(0x55555555), C.c_uint32(0x66666666))
    assert [int(x) for x in v] == [0xCCCCCCCB, 0x45B64444, 0x06FFFFFF, 0x07000000]


def test_blake3_compression_kat(oracle):
    """/root/reference/src/test_circuits/blake3.rs:2646-2746 (32-word state_in -> 16-word state_out)"""
    state_in = [0x1111 * i for i in range(16)] + [0x11110000 * i for i in range(16)]
    expected = [0xD304E51C, 0xC2DF34A0, 0x5EBA7F1F, 0x2AB9650F, 0xD9CEF159, 0x4E9D3A6A, 0xCAC2E310, 0xC6B9BE7E,
                0xAD9FD58A, 0x0899E71B, 0xCA51A599, 0xC3FBD7C0, 0x751D2F26, 0x6CD0AC6B, 0xC58F3C1D, 0xE6D65414]
    si = np.array(state_in, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    oracle.lib().mso_b3_rounds_kat(si.ctypes.data_as(C.POINTER(C.c_uint32)), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert [int(x) for x in out] == expected


def test_blake3_official_vectors(oracle):
    """Published BLAKE3 test vectors (input byte i = i % 251) for lengths across the chunk/tree boundaries."""
    vec = {
        0: "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262",
        1: "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213",
        1024: "42214739f095a406f3fc83deb889744ac00df831c10daa55189b5d121c855af7",
        1025: "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444",
    }
    for n, h in vec.items():
        assert oracle.hash_bytes(bytes(i % 251 for i in range(n))).hex() == h
    assert oracle.hash_bytes(b"abc").hex() == "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"


def test_serializing_hasher_and_compress(oracle):
    """SerializingHasher = BLAKE3 over 8 LE bytes per element; compress = BLAKE3(left || right) (src/types.rs:82-83,199)."""
    row = list(range(1, 18))
    assert oracle.hash_elems(row) == oracle.hash_bytes(b"".join(int(x).to_bytes(8, "little") for x in row))
    l, r = bytes(range(32)), bytes(range(32, 64))
    assert oracle.compress2(l, r) == oracle.hash_bytes(l + r)


def test_challenger_semantics(oracle):
    """HashChallenger: sample = hash(input) then pop from the BACK; u64 from 8 popped bytes little-endian."""
    seed = b"multi-stark/v0" + bytes(56)
    ch = oracle.Challenger(seed)
    ch.observe(0x0102030405060708)
    digest = oracle.hash_bytes(seed + (0x0102030405060708).to_bytes(8, "little"))
    want = int.from_bytes(bytes(digest[31 - k] for k in range(8)), "little") & ((1 << 20) - 1)
    assert ch.sample_bits(20) == want
    # after a flush the input buffer is the digest: observing then sampling hashes digest || new bytes
    ch.observe(5)
    d2 = oracle.hash_bytes(digest + (5).to_bytes(8, "little"))
    c0 = int.from_bytes(bytes(d2[31 - k] for k in range(8)), "little")
    c1 = int.from_bytes(bytes(d2[23 - k] for k in range(8)), "little")
    if c0 < P and c1 < P:
        assert ch.sample_ext() == (c0, c1)


def test_grind_returns_minimal_witness(oracle):
    ch = oracle.Challenger(b"seed")
    ch.observe(42)
    state = b"seed" + (42).to_bytes(8, "little")
    bits = 6
    w = ch.grind(bits)

    def ok(c):
        d = oracle.hash_bytes(state + c.to_bytes(8, "little"))
        return int.from_bytes(bytes(d[31 - k] for k in range(8)), "little") & ((1 << bits) - 1) == 0

    assert ok(w) and not any(ok(c) for c in range(w))
    assert oracle.Challenger(b"x").grind(0) == 0
