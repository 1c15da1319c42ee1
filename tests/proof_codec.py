"""Proof::to_bytes <-> Python structure for GoldilocksBlake3Config (the layout of src/prover.rs:201-248 as restated in
oracle/oracle_core.cpp proof_to_bytes): lets the tests tamper with a proof FIELD-wise the way the reference's verifier
tests do (src/verifier.rs:852-912: `proof.stage_1_opened_values[0][0][0] += ONE`, `proof.log_degrees.pop()`, ...).
`parse(b, elem_bytes=4, ext_degree=4)` reads the BabyBear / Poseidon2 configuration's proofs (elements = Montgomery words).
Test infrastructure only."""
import struct

_ELEM, _DEG = 8, 2  # set by parse / serialize


class _R:
    def __init__(self, b):
        self.b, self.o = b, 0

    def u8(self):
        v = self.b[self.o]
        self.o += 1
        return v

    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.o)[0]
        self.o += 8
        return v

    def cnt(self):
        """a length prefix: never more items than bytes left (a foreign or corrupted layout must fail, not loop)"""
        v = self.u64()
        if v > len(self.b) - self.o:
            raise ValueError("proof layout: a length prefix of %d at byte %d exceeds the %d bytes left" % (v, self.o - 8, len(self.b) - self.o))
        return v

    def fe(self):
        if self.o + _ELEM > len(self.b):
            raise ValueError("proof layout: truncated at byte %d" % self.o)
        v = int.from_bytes(self.b[self.o:self.o + _ELEM], "little")
        self.o += _ELEM
        return v

    def raw(self, n):
        v = bytes(self.b[self.o:self.o + n])
        self.o += n
        return v

    def ext(self):
        return [self.fe() for _ in range(_DEG)]

    def cap(self):
        return [self.raw(32) for _ in range(self.cnt())]

    def round(self):
        return [[[self.ext() for _ in range(self.cnt())] for _ in range(self.cnt())] for _ in range(self.cnt())]


class _W:
    def __init__(self):
        self.p = []

    def u8(self, v):
        self.p.append(struct.pack("<B", v))

    def u64(self, v):
        self.p.append(struct.pack("<Q", v))

    def fe(self, v):
        self.p.append(int(v).to_bytes(_ELEM, "little"))

    def ext(self, e):
        for c in e:
            self.fe(c)

    def cap(self, c):
        self.u64(len(c))
        self.p.extend(c)

    def round(self, r):
        self.u64(len(r))
        for m in r:
            self.u64(len(m))
            for pt in m:
                self.u64(len(pt))
                for e in pt:
                    self.ext(e)


def parse(b, elem_bytes=8, ext_degree=2):
    global _ELEM, _DEG
    _ELEM, _DEG = elem_bytes, ext_degree
    r = _R(b)
    p = {"active": [r.u8() for _ in range(r.cnt())]}
    p["stage_1_commit"], p["stage_2_commit"], p["quotient_commit"] = r.cap(), r.cap(), r.cap()
    p["intermediate_accumulators"] = [r.ext() for _ in range(r.cnt())]
    p["log_degrees"] = [r.u8() for _ in range(r.cnt())]
    fri = {"commit_phase_commits": [r.cap() for _ in range(r.cnt())], "commit_pow_witnesses": [r.fe() for _ in range(r.cnt())]}
    qs = []
    for _ in range(r.cnt()):
        q = {"input_proof": [], "commit_phase_openings": []}
        for _ in range(r.cnt()):
            rows = [[r.fe() for _ in range(r.cnt())] for _ in range(r.cnt())]
            q["input_proof"].append({"opened_values": rows, "proof": [r.raw(32) for _ in range(r.cnt())]})
        for _ in range(r.cnt()):
            la = r.u8()
            sib = [r.ext() for _ in range(r.cnt())]
            q["commit_phase_openings"].append({"log_arity": la, "sibling_values": sib, "proof": [r.raw(32) for _ in range(r.cnt())]})
        qs.append(q)
    fri["query_proofs"] = qs
    fri["final_poly"] = [r.ext() for _ in range(r.cnt())]
    fri["query_pow_witness"] = r.fe()
    p["opening_proof"] = fri
    p["quotient_opened_values"] = r.round()
    p["preprocessed_opened_values"] = r.round() if r.u8() else None
    p["stage_1_opened_values"] = r.round()
    p["stage_2_opened_values"] = r.round()
    assert r.o == len(b), "trailing bytes"
    return p


def serialize(p, elem_bytes=8, ext_degree=2):
    global _ELEM, _DEG
    _ELEM, _DEG = elem_bytes, ext_degree
    w = _W()
    w.u64(len(p["active"]))
    for a in p["active"]:
        w.u8(a)
    w.cap(p["stage_1_commit"]), w.cap(p["stage_2_commit"]), w.cap(p["quotient_commit"])
    w.u64(len(p["intermediate_accumulators"]))
    for e in p["intermediate_accumulators"]:
        w.ext(e)
    w.u64(len(p["log_degrees"]))
    for d in p["log_degrees"]:
        w.u8(d)
    f = p["opening_proof"]
    w.u64(len(f["commit_phase_commits"]))
    for c in f["commit_phase_commits"]:
        w.cap(c)
    w.u64(len(f["commit_pow_witnesses"]))
    for x in f["commit_pow_witnesses"]:
        w.fe(x)
    w.u64(len(f["query_proofs"]))
    for q in f["query_proofs"]:
        w.u64(len(q["input_proof"]))
        for bo in q["input_proof"]:
            w.u64(len(bo["opened_values"]))
            for row in bo["opened_values"]:
                w.u64(len(row))
                for x in row:
                    w.fe(x)
            w.u64(len(bo["proof"]))
            w.p.extend(bo["proof"])
        w.u64(len(q["commit_phase_openings"]))
        for s in q["commit_phase_openings"]:
            w.u8(s["log_arity"])
            w.u64(len(s["sibling_values"]))
            for e in s["sibling_values"]:
                w.ext(e)
            w.u64(len(s["proof"]))
            w.p.extend(s["proof"])
    w.u64(len(f["final_poly"]))
    for e in f["final_poly"]:
        w.ext(e)
    w.fe(f["query_pow_witness"])
    w.round(p["quotient_opened_values"])
    if p["preprocessed_opened_values"] is None:
        w.u8(0)
    else:
        w.u8(1)
        w.round(p["preprocessed_opened_values"])
    w.round(p["stage_1_opened_values"])
    w.round(p["stage_2_opened_values"])
    return b"".join(w.p)
