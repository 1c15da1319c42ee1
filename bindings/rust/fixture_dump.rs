//! Reference-side fixture generator: turns every proof the multi-stark test suite produces into one line of JSON that
//! pins the MI355X build's oracle (and, through it, the HIP prover) against the real reference — commitments,
//! challenges, FRI contents and `Proof::to_bytes`, byte for byte.
//!
//! NOT COMPILED where this repository was built (no rustc / cargo there). It is written against the reference at the
//! revision surveyed (argumentcomputer/multi-stark with Plonky3 rev e9d75614, `Cargo.toml:17-30`) and uses only items that
//! are `pub` in that tree: `System::{config, circuits, preprocessed_commit}` (`src/system.rs:85-101`), `Circuit` fields
//! (`:49-71`), `ConstraintGraph::{nodes, zeros, lookups}` (`src/graph.rs:62-76`), `Node` / `ColRef` / `Source` /
//! `RowOffset` (`src/graph.rs:35-46`, `src/expr.rs:13-35`), `Lookup::{multiplicity, args}` (`src/lookup.rs:37-40`),
//! `SystemWitness::traces` (`src/system.rs:229`), `Proof::to_bytes` (`src/prover.rs:245`).
//!
//! How to use it ("Pinning the oracle", has the same steps):
//!   1. copy this file to `src/fixture_dump.rs` of the reference checkout;
//!   2. `git apply --unidiff-zero bindings/rust/fixture_hook.patch` — three one-line insertions: `#[cfg(test)] mod
//!      fixture_dump;` in `src/lib.rs`, and in `System::prove_multiple_claims` (`src/prover.rs:290-603`) a clone of the
//!      witness traces at the top and a call of [`case`] on the finished proof (both `#[cfg(test)]`);
//!   3. `MSTARK_FIXTURES=$PWD/reference_refs.jsonl cargo test --release` (NOT with `--test-threads=1`: the harness then
//!      runs every test on the thread "main" and the cases lose their names; and WITHOUT `--features parallel`: p3's `grind`
//!      returns whichever witness a rayon thread finds first, `src/types.rs:31-42`, the serial build returns the smallest,
//!      which is what the MI355X prover is pinned to);
//!   4. drop `reference_refs.jsonl` into `tests/golden/` of the MI355X repository and run
//!      `python -m pytest tests/test_reference_pins.py`: parity flips from "unpinned" to checked, field by field.
//! Every test of the suite that proves something then contributes a case named after the test (the harness names the
//! thread): `u32_add_proof` (`src/test_circuits/u32_add.rs:193-221`), `lookup_test` (`src/lookup.rs:1043-1051`),
//! `baby_bear_poseidon2_smoke_test` (`src/test_circuits/baby_bear_config.rs:159-206`), the Blake3 circuit
//! (`src/test_circuits/blake3.rs:2215-2613`), `byte_test`, the verifier's cases ... plus the tests at the end of this file:
//! the `gen_pcs_refs` / `gen_challenger_refs` scenarios of `src/types.rs:246-319` as JSON, `examples/simple_proof.rs` at
//! 4 and 4096 rows, under the bench parameters (`benches/multi_stark.rs:244-258`: the proof-of-work path) and with
//! `max_log_arity` 2 / 3 (FRI rounds of arity 4 / 8), and the
//! Poseidon2 constants of the BabyBear configuration.
//!
//! A field element is written as its "serde word": the little-endian integer of the bytes `serde` gives it under the
//! crate's own bincode configuration (`src/prover.rs:241-243`) — the canonical u64 for Goldilocks, the 32-bit Montgomery
//! word for BabyBear — i.e. exactly what `Proof::to_bytes` contains. `elem_bytes` says which.
#![allow(dead_code, clippy::cast_possible_truncation)]

use std::fmt::Write as _;
use std::io::Write as _;
use std::sync::Mutex;

use p3_matrix::Matrix;
use p3_matrix::dense::RowMajorMatrix;
use serde::Serialize;

use crate::config::{Com, StarkGenericConfig, Val};
use crate::expr::{RowOffset, Source};
use crate::graph::Node;
use crate::prover::Proof;
use crate::system::System;

static OUT: Mutex<()> = Mutex::new(());

fn out_path() -> String {
    std::env::var("MSTARK_FIXTURES").unwrap_or_else(|_| "target/reference_refs.jsonl".to_string())
}

/// One JSON object per line, appended under a lock (the harness runs tests on several threads unless told otherwise).
fn emit(line: &str) {
    let _guard = OUT.lock().unwrap_or_else(|e| e.into_inner());
    let mut f = std::fs::OpenOptions::new()
        .create(true)
        .append(true)
        .open(out_path())
        .expect("cannot open the fixture file (MSTARK_FIXTURES)");
    writeln!(f, "{line}").expect("cannot write the fixture file");
}

fn bincode_bytes<T: Serialize>(x: &T) -> Vec<u8> {
    let cfg = bincode::config::standard()
        .with_little_endian()
        .with_fixed_int_encoding();
    bincode::serde::encode_to_vec(x, cfg).expect("serde of a fixture value failed")
}

/// (serde word, number of bytes) of one field element.
fn word<T: Serialize>(x: &T) -> (u64, usize) {
    let b = bincode_bytes(x);
    assert!(b.len() <= 8, "field element wider than 8 bytes");
    let mut v = 0u64;
    for (i, byte) in b.iter().enumerate() {
        v |= u64::from(*byte) << (8 * i);
    }
    (v, b.len())
}

fn hex(bytes: &[u8]) -> String {
    let mut s = String::with_capacity(2 * bytes.len());
    for b in bytes {
        write!(s, "{b:02x}").unwrap();
    }
    s
}

fn words<T: Serialize>(xs: impl IntoIterator<Item = T>) -> String {
    let mut s = String::from("[");
    for (i, x) in xs.into_iter().enumerate() {
        if i > 0 {
            s.push(',');
        }
        write!(s, "{}", word(&x).0).unwrap();
    }
    s.push(']');
    s
}

fn test_name() -> String {
    std::thread::current()
        .name()
        .unwrap_or("unnamed")
        .replace('"', "'")
}

/// The hook `prove_multiple_claims` calls on its finished proof (see the module docs): system, claims, the stage-1
/// traces it was given and the proof. Traces above `MSTARK_FIXTURE_MAX_WORDS` (default 2^24 elements in total) are
/// left out; the case then still pins the verifier side (the oracle's verifier must accept `proof_hex`).
pub(crate) fn case<SC: StarkGenericConfig>(
    system: &System<SC>,
    claims: &[&[Val<SC>]],
    traces: &[RowMajorMatrix<Val<SC>>],
    proof: &Proof<SC>,
) {
    // (Com<SC>: Serialize comes with p3_commit::Pcs::Commitment, Val<SC>: Serialize with p3_field::Field)
    let mut s = String::new();
    let elem_bytes = system
        .circuits
        .iter()
        .flat_map(|c| c.graph.nodes.iter())
        .find_map(|n| match n {
            Node::Const(x) => Some(word(x).1),
            _ => None,
        })
        .or_else(|| traces.iter().flat_map(|t| t.values.first()).map(|x| word(x).1).next())
        .unwrap_or(8);
    write!(
        s,
        "{{\"kind\":\"proof\",\"test\":\"{}\",\"config\":\"{}\",\"log_blowup\":{},\"elem_bytes\":{},\"circuits\":[",
        test_name(),
        std::any::type_name::<SC>(),
        system.config.log_blowup(),
        elem_bytes
    )
    .unwrap();
    for (ci, c) in system.circuits.iter().enumerate() {
        if ci > 0 {
            s.push(',');
        }
        write!(
            s,
            "{{\"main_width\":{},\"preprocessed_width\":{},\"preprocessed_height\":{},\"num_lookups\":{},\"stage_2_width\":{},\
             \"constraint_count\":{},\"max_constraint_degree\":{},\"lookup_prefix_len\":{},\"nodes\":[",
            c.main_width,
            c.preprocessed_width,
            c.preprocessed_height,
            c.num_lookups,
            c.stage_2_width,
            c.constraint_count,
            c.max_constraint_degree,
            c.graph.lookup_prefix_len
        )
        .unwrap();
        // node = [kind, a, b, source, offset]; kinds in the declaration order of graph::Node (src/graph.rs:35-46):
        // 0 Const(a = serde word) 1 Var(a = column, source 0 preprocessed / 1 main / 2 stage-2, offset 0 current / 1 next)
        // 2 Public(a) 3 IsFirstRow 4 IsLastRow 5 IsTransition 6 Add(a, b) 7 Sub(a, b) 8 Mul(a, b) 9 Neg(a)
        for (i, n) in c.graph.nodes.iter().enumerate() {
            if i > 0 {
                s.push(',');
            }
            let (kind, a, b, src, off): (u32, u64, u64, u32, u32) = match n {
                Node::Const(x) => (0, word(x).0, 0, 0, 0),
                Node::Var(col) => (
                    1,
                    u64::from(col.index),
                    0,
                    match col.source {
                        Source::Preprocessed => 0,
                        Source::Main => 1,
                        Source::Stage2 => 2,
                    },
                    match col.offset {
                        RowOffset::Current => 0,
                        RowOffset::Next => 1,
                    },
                ),
                Node::Public(i) => (2, u64::from(*i), 0, 0, 0),
                Node::IsFirstRow => (3, 0, 0, 0, 0),
                Node::IsLastRow => (4, 0, 0, 0, 0),
                Node::IsTransition => (5, 0, 0, 0, 0),
                Node::Add(x, y) => (6, u64::from(x.0), u64::from(y.0), 0, 0),
                Node::Sub(x, y) => (7, u64::from(x.0), u64::from(y.0), 0, 0),
                Node::Mul(x, y) => (8, u64::from(x.0), u64::from(y.0), 0, 0),
                Node::Neg(x) => (9, u64::from(x.0), 0, 0, 0),
            };
            write!(s, "[{kind},{a},{b},{src},{off}]").unwrap();
        }
        s.push_str("],\"zeros\":[");
        for (i, z) in c.graph.zeros.iter().enumerate() {
            if i > 0 {
                s.push(',');
            }
            write!(s, "{}", z.0).unwrap();
        }
        s.push_str("],\"lookups\":[");
        for (i, l) in c.graph.lookups.iter().enumerate() {
            if i > 0 {
                s.push(',');
            }
            write!(s, "[{},[", l.multiplicity.0).unwrap();
            for (k, a) in l.args.iter().enumerate() {
                if k > 0 {
                    s.push(',');
                }
                write!(s, "{}", a.0).unwrap();
            }
            s.push_str("]]");
        }
        s.push_str("],\"preprocessed\":");
        match &c.preprocessed {
            Some(m) => s.push_str(&words(m.values.iter().copied())),
            None => s.push_str("null"),
        }
        s.push('}');
    }
    s.push_str("],\"preprocessed_commit_hex\":");
    match &system.preprocessed_commit {
        Some(c) => write!(s, "\"{}\"", hex(&bincode_bytes(c))).unwrap(),
        None => s.push_str("null"),
    }
    let max_words: usize = std::env::var("MSTARK_FIXTURE_MAX_WORDS")
        .ok()
        .and_then(|v| v.parse().ok())
        .unwrap_or(1 << 24);
    let total: usize = traces.iter().map(|t| t.values.len()).sum();
    s.push_str(",\"traces\":");
    if total <= max_words {
        s.push('[');
        for (i, t) in traces.iter().enumerate() {
            if i > 0 {
                s.push(',');
            }
            write!(s, "{{\"height\":{},\"width\":{},\"values\":{}}}", t.height(), t.width(), words(t.values.iter().copied())).unwrap();
        }
        s.push(']');
    } else {
        s.push_str("null");
    }
    s.push_str(",\"trace_heights\":[");
    for (i, t) in traces.iter().enumerate() {
        if i > 0 {
            s.push(',');
        }
        write!(s, "{}", t.height()).unwrap();
    }
    s.push_str("],\"claims\":[");
    for (i, c) in claims.iter().enumerate() {
        if i > 0 {
            s.push(',');
        }
        s.push_str(&words(c.iter().copied()));
    }
    let bytes = proof.to_bytes().expect("Proof::to_bytes failed");
    write!(s, "],\"proof_len\":{},\"proof_hex\":\"{}\"}}", bytes.len(), hex(&bytes)).unwrap();
    emit(&s);
}

#[cfg(test)]
mod tests {
    use super::*;
    use crate::p3_adapter::LookupAir;
    use crate::system::SystemWitness;
    use crate::types::{
        Challenger, CommitmentParameters, ExtVal, FriParameters, GoldilocksBlake3Config, Mmcs, Val as GVal,
    };
    use p3_air::{Air, AirBuilder, BaseAir, WindowAccess};
    use p3_blake3::Blake3;
    use p3_challenger::{CanObserve, CanSampleBits, FieldChallenger};
    use p3_commit::Mmcs as _;
    use p3_field::{BasedVectorSpace, PrimeCharacteristicRing, PrimeField64};
    use p3_symmetric::{
        CompressionFunctionFromHasher, CryptographicHasher, PseudoCompressionFunction, SerializingHasher,
    };

    fn limbs(d: [u8; 32]) -> [u64; 4] {
        core::array::from_fn(|i| u64::from_le_bytes(d[i * 8..i * 8 + 8].try_into().unwrap()))
    }
    fn dig(xs: [u64; 4]) -> [u8; 32] {
        let mut o = [0u8; 32];
        for i in 0..4 {
            o[i * 8..i * 8 + 8].copy_from_slice(&xs[i].to_le_bytes());
        }
        o
    }

    /// The scenarios of `gen_pcs_refs` (`src/types.rs:246-285`) as one JSON line instead of prints.
    #[test]
    fn pcs_refs_json() {
        let f = GVal::from_u32;
        let fh = SerializingHasher::new(Blake3);
        let mut s = String::from("{\"kind\":\"pcs_refs\"");
        for n in [3u32, 17, 22, 20] {
            let row: Vec<GVal> = (1..=n).map(f).collect();
            let d: [u8; 32] = fh.hash_iter(row);
            write!(s, ",\"LEAF{}\":{:?}", n, limbs(d)).unwrap();
        }
        let comp = CompressionFunctionFromHasher::<Blake3, 2, 32>::new(Blake3);
        let c: [u8; 32] = comp.compress([dig([1, 2, 3, 4]), dig([5, 6, 7, 8])]);
        write!(s, ",\"COMPRESS\":{:?}", limbs(c)).unwrap();
        // Merkle tree: heights 8/4/2, widths 2/3/1, opened at index 5 (src/types.rs:262-283)
        let mut m0 = vec![f(0); 16];
        m0[10] = f(11);
        m0[11] = f(12);
        let mut m1 = vec![f(0); 12];
        m1[6] = f(107);
        m1[7] = f(108);
        m1[8] = f(109);
        let mut m2 = vec![f(0); 2];
        m2[1] = f(202);
        let mmcs = Mmcs::new(SerializingHasher::new(Blake3), CompressionFunctionFromHasher::<Blake3, 2, 32>::new(Blake3), 0);
        let (commit, pd) = mmcs.commit(vec![
            RowMajorMatrix::new(m0, 2),
            RowMajorMatrix::new(m1, 3),
            RowMajorMatrix::new(m2, 1),
        ]);
        let bo = mmcs.open_batch(5, &pd);
        s.push_str(",\"OPENED\":[");
        let mut first = true;
        for row in &bo.opened_values {
            for v in row {
                if !first {
                    s.push(',');
                }
                first = false;
                write!(s, "{}", v.as_canonical_u64()).unwrap();
            }
        }
        s.push_str("],\"SIBLINGS\":[");
        for (i, sib) in bo.opening_proof.iter().enumerate() {
            if i > 0 {
                s.push(',');
            }
            write!(s, "{:?}", limbs(*sib)).unwrap();
        }
        write!(s, "],\"COMMIT_hex\":\"{}\"}}", hex(&bincode_bytes(&commit))).unwrap();
        emit(&s);
    }

    /// The scenarios of `gen_challenger_refs` (`src/types.rs:287-318`) as one JSON line.
    #[test]
    fn challenger_refs_json() {
        let g = GVal::from_u64;
        fn el(e: ExtVal) -> (u64, u64) {
            let c: &[GVal] = e.as_basis_coefficients_slice();
            (c[0].as_canonical_u64(), c[1].as_canonical_u64())
        }
        let mut ch = Challenger::from_hasher(vec![], Blake3);
        ch.observe(g(0x0102030405060708));
        let sb: usize = CanSampleBits::<usize>::sample_bits(&mut ch, 20);
        let mut ch = Challenger::from_hasher(vec![], Blake3);
        ch.observe(g(0x0102030405060708));
        ch.observe(g(0x1122334455667788));
        let apcs: ExtVal = ch.sample_algebra_element();
        let afri: ExtVal = ch.sample_algebra_element();
        ch.observe(g(0x00000000deadbeef));
        let beta: ExtVal = ch.sample_algebra_element();
        ch.observe(g(0x0a0b0c0d01020304));
        ch.observe(g(0x0000000000000002));
        let sb2: usize = CanSampleBits::<usize>::sample_bits(&mut ch, 20);
        let (a, b, c) = (el(apcs), el(afri), el(beta));
        emit(&format!(
            "{{\"kind\":\"challenger_refs\",\"SAMPLE_BITS\":{sb},\"APCS\":[{},{}],\"AFRI\":[{},{}],\"BETA\":[{},{}],\"SAMPLE_BITS2\":{sb2}}}",
            a.0, a.1, b.0, b.1, c.0, c.1
        ));
    }

    /// a^2 + b^2 = c^2 on three columns: the AIR of `examples/simple_proof.rs:21-44` (an example cannot be reached from a
    /// unit test, so the three lines are restated here).
    struct PythagoreanAir;
    impl<F> BaseAir<F> for PythagoreanAir {
        fn width(&self) -> usize {
            3
        }
    }
    impl<AB: AirBuilder> Air<AB> for PythagoreanAir
    where
        AB::Var: Copy,
    {
        fn eval(&self, builder: &mut AB) {
            let main = builder.main();
            let row = main.current_slice();
            builder.assert_eq(row[0] * row[0] + row[1] * row[1], row[2] * row[2]);
        }
    }

    fn pythagorean(rows: usize, commitment: CommitmentParameters, fri: FriParameters) {
        let config = GoldilocksBlake3Config::new(commitment, fri);
        let (system, key) = System::new(config, [LookupAir::new(PythagoreanAir, vec![])]);
        let triples: [[u32; 3]; 4] = [[3, 4, 5], [5, 12, 13], [8, 15, 17], [7, 24, 25]];
        let values: Vec<GVal> = (0..rows)
            .flat_map(|r| triples[r % 4])
            .map(GVal::from_u32)
            .collect();
        let witness = SystemWitness::from_stage_1(vec![RowMajorMatrix::new(values, 3)], &system);
        let no_claims: &[&[GVal]] = &[];
        // the hook inside prove_multiple_claims writes the case
        let proof = system.prove_multiple_claims(&key, no_claims, witness);
        system.verify_multiple_claims(no_claims, &proof).unwrap();
    }

    const TEST_COMMITMENT: CommitmentParameters = CommitmentParameters { log_blowup: 1, cap_height: 0 };
    const TEST_FRI: FriParameters = FriParameters {
        log_final_poly_len: 0,
        max_log_arity: 1,
        num_queries: 64,
        commit_proof_of_work_bits: 0,
        query_proof_of_work_bits: 0,
    };

    /// `examples/simple_proof.rs:46-91` (4 rows) and BASELINE config 1 (4096 rows).
    #[test]
    fn simple_proof_4_rows() {
        pythagorean(4, TEST_COMMITMENT, TEST_FRI);
    }
    #[test]
    fn simple_proof_4096_rows() {
        pythagorean(4096, TEST_COMMITMENT, TEST_FRI);
    }
    /// The same AIR under `bench_config()` (`benches/multi_stark.rs:244-258`): log_blowup 2, 100 queries and 10 + 10
    /// proof-of-work bits, which no unit test of the crate exercises. Serial build only (see the module docs).
    #[test]
    fn simple_proof_bench_params() {
        pythagorean(
            1024,
            CommitmentParameters { log_blowup: 2, cap_height: 0 },
            FriParameters {
                log_final_poly_len: 0,
                max_log_arity: 1,
                num_queries: 100,
                commit_proof_of_work_bits: 10,
                query_proof_of_work_bits: 10,
            },
        );
    }
    /// Caps, a longer final polynomial and few queries: the shapes the other cases leave at their defaults.
    #[test]
    fn simple_proof_cap2_final4() {
        pythagorean(
            256,
            CommitmentParameters { log_blowup: 2, cap_height: 2 },
            FriParameters {
                log_final_poly_len: 2,
                max_log_arity: 1,
                num_queries: 20,
                commit_proof_of_work_bits: 3,
                query_proof_of_work_bits: 5,
            },
        );
    }

    /// `max_log_arity` above 1 (`src/types.rs:189-190`), which no call site of the crate sets: rounds of arity 4 and 8, the
    /// second with proof of work and a 2-coefficient final polynomial (the arity schedule has to stop at the final height).
    /// These pin the row layout of a wide round, the schedule and the roll-in factor of the repository's restatement.
    #[test]
    fn simple_proof_arity4() {
        pythagorean(
            256,
            CommitmentParameters { log_blowup: 1, cap_height: 0 },
            FriParameters {
                log_final_poly_len: 0,
                max_log_arity: 2,
                num_queries: 20,
                commit_proof_of_work_bits: 0,
                query_proof_of_work_bits: 0,
            },
        );
    }
    #[test]
    fn simple_proof_arity8_pow() {
        pythagorean(
            512,
            CommitmentParameters { log_blowup: 2, cap_height: 1 },
            FriParameters {
                log_final_poly_len: 1,
                max_log_arity: 3,
                num_queries: 20,
                commit_proof_of_work_bits: 3,
                query_proof_of_work_bits: 5,
            },
        );
    }

    /// The Poseidon2 constants of the BabyBear configuration: `Perm::new_from_rng_128(&mut SmallRng::seed_from_u64(42))`
    /// (`src/test_circuits/baby_bear_config.rs:54-55`). p3-poseidon2 keeps them private, so the stream is replayed the way
    /// `Poseidon2::new_from_rng` draws it (4 initial external rounds x 16, 4 terminal x 16, then 13 internal constants) and
    /// the REAL permutation's image of [0, 1, .., 15] is written next to them: the consumer rebuilds the permutation from
    /// the constants and must reproduce that image, which tells a wrong replay from a wrong oracle.
    #[test]
    fn babybear_poseidon2_constants() {
        use p3_baby_bear::{BabyBear, Poseidon2BabyBear};
        use p3_symmetric::Permutation;
        use rand::distr::StandardUniform;
        use rand::rngs::SmallRng;
        use rand::{RngExt, SeedableRng}; // rand 0.10 (Cargo.toml:36): the sampling methods live on RngExt, as in src/prover.rs:968

        let mut rng = SmallRng::seed_from_u64(42);
        let perm = Poseidon2BabyBear::<16>::new_from_rng_128(&mut rng);
        let mut state: [BabyBear; 16] = core::array::from_fn(|i| BabyBear::from_u32(i as u32));
        perm.permute_mut(&mut state);

        let mut rng = SmallRng::seed_from_u64(42);
        let initial: Vec<[BabyBear; 16]> = (&mut rng).sample_iter(StandardUniform).take(4).collect();
        let terminal: Vec<[BabyBear; 16]> = (&mut rng).sample_iter(StandardUniform).take(4).collect();
        let internal: Vec<BabyBear> = (&mut rng).sample_iter(StandardUniform).take(13).collect();
        let canon = |x: &BabyBear| x.as_canonical_u64();
        let mut s = String::from("{\"kind\":\"babybear_poseidon2\",\"external_initial\":[");
        let flat = |rounds: &Vec<[BabyBear; 16]>| rounds.iter().flatten().map(canon).map(|v| v.to_string()).collect::<Vec<_>>().join(",");
        s.push_str(&flat(&initial));
        s.push_str("],\"external_terminal\":[");
        s.push_str(&flat(&terminal));
        s.push_str("],\"internal\":[");
        s.push_str(&internal.iter().map(canon).map(|v| v.to_string()).collect::<Vec<_>>().join(","));
        s.push_str("],\"permute_0_to_15\":[");
        s.push_str(&state.iter().map(canon).map(|v| v.to_string()).collect::<Vec<_>>().join(","));
        s.push_str("]}");
        emit(&s);
    }
}
