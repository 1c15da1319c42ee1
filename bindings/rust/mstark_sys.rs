//! Raw FFI declarations for `libmstark_hip.so` — one line per entry point of `include/mstark.h` and
//! `include/mstark_bb.h`, for a maintainer of argumentcomputer/multi-stark who wants to call the MI355X prover from
//! the crate (see INTEGRATION.md for the call sites it replaces in `src/prover.rs`).
//!
//! NOT COMPILED in the environment this repository was built in (no rustc / cargo there): it is source for the
//! reference side, kept in step with the headers by `tests/test_abi.py::test_rust_binding_lists_every_symbol`.
//! Link with `println!("cargo:rustc-link-lib=dylib=mstark_hip");` in `build.rs`.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_void};

#[repr(C)] pub struct ms_ctx { _p: [u8; 0] }
#[repr(C)] pub struct ms_system { _p: [u8; 0] }
#[repr(C)] pub struct ms_witness { _p: [u8; 0] }
#[repr(C)] pub struct ms_mmcs { _p: [u8; 0] }
#[repr(C)] pub struct ms_challenger { _p: [u8; 0] }
#[repr(C)] pub struct ms_comm_rccl { _p: [u8; 0] }
#[repr(C)] pub struct ms_comm_local_group { _p: [u8; 0] }
#[repr(C)] pub struct ms_comm_local { _p: [u8; 0] }
#[repr(C)] pub struct ms_trace { _p: [u8; 0] }
#[repr(C)] pub struct msbb_system { _p: [u8; 0] }
#[repr(C)] pub struct msbb_witness { _p: [u8; 0] }
#[repr(C)] pub struct msbb_mmcs { _p: [u8; 0] }
#[repr(C)] pub struct msbb_challenger { _p: [u8; 0] }
#[repr(C)] pub struct msbb_trace { _p: [u8; 0] }

pub const MS_OK: i32 = 0;
pub const MS_ERR: i32 = -1;
pub const MS_ERR_NO_DEVICE: i32 = -2;
pub const MS_ERR_BUFFER: i32 = -3;
/// `verdict` of ms_verify / msbb_verify: 0 = Ok(()), else the VerificationError variant (src/verifier.rs:176-192)
pub const MS_VERDICT_ACCEPT: i32 = 0;
pub const MS_VERDICT_INVALID_OPENING: i32 = 2;
pub const MS_VERDICT_INVALID_SHAPE: i32 = 3;
pub const MS_VERDICT_INVALID_SYSTEM: i32 = 4;
pub const MS_VERDICT_OOD_MISMATCH: i32 = 5;
pub const MS_VERDICT_UNBALANCED: i32 = 6;

/// Exchanges `ms_prove_sharded` calls back for (device pointers of the context's device; 0 = ok).
#[repr(C)]
pub struct ms_comm {
    /// `core::mem::size_of::<ms_comm>() as u32`: the library treats members beyond it as not offered
    pub size: u32,
    pub rank: i32,
    pub world: i32,
    pub user: *mut c_void,
    pub all_to_all: Option<unsafe extern "C" fn(user: *mut c_void, send_dev: *const c_void, recv_dev: *mut c_void, bytes_per_peer: usize) -> i32>,
    pub all_gather: Option<unsafe extern "C" fn(user: *mut c_void, send_dev: *const c_void, recv_dev: *mut c_void, bytes: usize) -> i32>,
    pub all_to_all_start: Option<unsafe extern "C" fn(user: *mut c_void, send_dev: *const c_void, send_stride: usize, recv_dev: *mut c_void,
                                                      recv_stride: usize, bytes_per_peer: usize) -> i32>,
    pub all_to_all_wait: Option<unsafe extern "C" fn(user: *mut c_void) -> i32>,
    pub all_to_all_cols_start: Option<unsafe extern "C" fn(user: *mut c_void, send_dev: *const c_void, send_peer_stride: usize, send_col_stride: usize,
                                                           recv_dev: *mut c_void, recv_peer_stride: usize, recv_col_stride: usize, ncols: usize,
                                                           seg_bytes: usize) -> i32>,
    pub set_stream_ordered: Option<unsafe extern "C" fn(user: *mut c_void, hip_stream: *mut c_void) -> i32>,
    pub all_to_all_cols_start2: Option<unsafe extern "C" fn(user: *mut c_void, send_dev: *const c_void, send_peer_stride: usize, send_col_stride: usize,
                                                            recv_dev: *mut c_void, recv_peer_stride: usize, recv_col_stride: usize, ncols: usize,
                                                            seg_bytes: usize, flags: u32) -> i32>,
    pub scatter_cols_start: Option<unsafe extern "C" fn(user: *mut c_void, root: i32, send_dev: *const c_void, send_peer_stride: usize,
                                                        send_col_stride: usize, recv_dev: *mut c_void, recv_col_stride: usize, ncols: usize,
                                                        seg_bytes: usize) -> i32>,
    pub abort: Option<unsafe extern "C" fn(user: *mut c_void, why: *const c_char)>,
}
pub const MS_COMM_SKIP_SELF: u32 = 1;

extern "C" {
    // ---- include/mstark.h (GoldilocksBlake3Config)
    pub fn ms_last_error() -> *const c_char;
    pub fn ms_device_count() -> i32;
    pub fn ms_ctx_create(device: i32, out: *mut *mut ms_ctx) -> i32;
    pub fn ms_ctx_destroy(ctx: *mut ms_ctx);
    pub fn ms_ctx_sync(ctx: *mut ms_ctx) -> i32;
    pub fn ms_ctx_sync_count(ctx: *mut ms_ctx, out: *mut u64) -> i32;
    pub fn ms_ctx_trim(ctx: *mut ms_ctx) -> i32;
    pub fn ms_ctx_set_profile_mask(ctx: *mut ms_ctx, mask: u32) -> i32;
    pub fn ms_ctx_kernel_stats(ctx: *mut ms_ctx, kernel_id: i32, launches: *mut u64, ms: *mut f64, alg_bytes: *mut f64) -> i32;
    pub fn ms_ctx_reset_stats(ctx: *mut ms_ctx) -> i32;
    pub fn ms_ctx_kernel_units(ctx: *mut ms_ctx, kernel_id: i32, units: *mut f64) -> i32;
    pub fn ms_ctx_debug_fail_alloc(ctx: *mut ms_ctx, nth: i32) -> i32;
    pub fn ms_kernel_count() -> i32;
    pub fn ms_kernel_name(kernel_id: i32) -> *const c_char;
    pub fn ms_system_create(ctx: *mut ms_ctx, blob: *const u8, len: usize, out: *mut *mut ms_system) -> i32;
    pub fn ms_system_destroy(sys: *mut ms_system);
    pub fn ms_system_preprocessed_commit(sys: *const ms_system, out: *mut u8, cap: usize, n_digests: *mut usize) -> i32;
    pub fn ms_system_circuit_info(sys: *const ms_system, circuit: usize, out9: *mut u64) -> i32;
    pub fn ms_witness_create(sys: *mut ms_system, traces: *const *const u64, heights: *const u64, mult: *const *const u64,
                             args: *const *const u64, n_claims: usize, claim_offsets: *const u64, claim_data: *const u64,
                             out: *mut *mut ms_witness) -> i32;
    pub fn ms_witness_create_host(sys: *mut ms_system, traces: *const *const u64, heights: *const u64, n_claims: usize,
                                  claim_offsets: *const u64, claim_data: *const u64, pinned: *mut i32,
                                  out: *mut *mut ms_witness) -> i32;
    pub fn ms_claims_slice_range(sys: *mut ms_system, heights: *const u64, n_claims: usize, claim_offsets: *const u64, rank: i32, world: i32,
                                 first_elem: *mut u64, n_elems: *mut u64) -> i32;
    pub fn ms_witness_create_host_sliced(sys: *mut ms_system, traces: *const *const u64, heights: *const u64, n_claims: usize,
                                         claim_offsets: *const u64, data_first: u64, data_count: u64, data_slice: *const u64,
                                         head: *const u64, n_head: usize, pinned: *mut i32, out: *mut *mut ms_witness) -> i32;
    pub fn ms_witness_prefetch(w: *mut ms_witness, on: i32) -> i32;
    pub fn ms_witness_u32_add_bench(sys: *mut ms_system, num_adds: usize, a0: u32, b0: u32, out: *mut *mut ms_witness) -> i32;
    pub fn ms_witness_destroy(w: *mut ms_witness);
    pub fn ms_prove(sys: *mut ms_system, w: *mut ms_witness, proof_out: *mut u8, cap: usize, proof_len: *mut usize, stage_ms: *mut f64) -> i32;
    pub fn ms_verify(sys: *mut ms_system, n_claims: usize, claim_offsets: *const u64, claim_data: *const u64, proof: *const u8,
                     proof_len: usize, verdict: *mut i32) -> i32;
    pub fn ms_prove_sharded(sys: *mut ms_system, w: *mut ms_witness, comm: *const ms_comm, owners: *const i32, proof_out: *mut u8,
                            cap: usize, proof_len: *mut usize, stage_ms: *mut f64) -> i32;
    pub fn ms_ctx_comm_progress(ctx: *mut ms_ctx, out: *mut c_char, cap: usize, seq: *mut u64, in_flight: *mut i32) -> i32;
    pub fn ms_comm_rccl_unique_id(out: *mut u8) -> i32;
    pub fn ms_comm_rccl_create(ctx: *mut ms_ctx, unique_id: *const u8, rank: i32, world: i32, out: *mut *mut ms_comm_rccl) -> i32;
    pub fn ms_comm_rccl_table(c: *mut ms_comm_rccl) -> *const ms_comm;
    pub fn ms_comm_rccl_bytes_moved(c: *mut ms_comm_rccl) -> u64;
    pub fn ms_comm_rccl_destroy(c: *mut ms_comm_rccl);
    pub fn ms_comm_local_group_create(world: i32, out: *mut *mut ms_comm_local_group) -> i32;
    pub fn ms_comm_local_group_abort(g: *mut ms_comm_local_group);
    pub fn ms_comm_local_group_destroy(g: *mut ms_comm_local_group);
    pub fn ms_comm_local_create(g: *mut ms_comm_local_group, ctx: *mut ms_ctx, rank: i32, out: *mut *mut ms_comm_local) -> i32;
    pub fn ms_comm_local_table(c: *mut ms_comm_local) -> *const ms_comm;
    pub fn ms_comm_local_bytes_moved(c: *mut ms_comm_local) -> u64;
    pub fn ms_comm_local_destroy(c: *mut ms_comm_local);
    pub fn ms_dft_batch(ctx: *mut ms_ctx, input: *const u64, h: usize, w: usize, inverse: i32, out: *mut u64) -> i32;
    pub fn ms_coset_lde_batch(ctx: *mut ms_ctx, input: *const u64, h: usize, w: usize, log_blowup: u32, out: *mut u64) -> i32;
    pub fn ms_quotient_lde(ctx: *mut ms_ctx, input: *const u64, log_n: u32, log_q: u32, log_blowup: u32, d: usize, out: *mut u64) -> i32;
    pub fn ms_mmcs_commit(ctx: *mut ms_ctx, n: usize, mats: *const *const u64, heights: *const u64, widths: *const u64, cap_height: u32,
                          cap_out: *mut u8, out: *mut *mut ms_mmcs) -> i32;
    pub fn ms_mmcs_open(m: *mut ms_mmcs, index: usize, vals_out: *mut u64, proof_out: *mut u8, n_siblings: *mut usize) -> i32;
    pub fn ms_mmcs_destroy(m: *mut ms_mmcs);
    pub fn ms_blake3(ctx: *mut ms_ctx, bytes: *const u8, len: usize, out32: *mut u8) -> i32;
    pub fn ms_challenger_create(params7: *const u64, out: *mut *mut ms_challenger) -> i32;
    pub fn ms_challenger_destroy(ch: *mut ms_challenger);
    pub fn ms_challenger_observe(ch: *mut ms_challenger, elems: *const u64, n: usize) -> i32;
    pub fn ms_challenger_observe_digests(ch: *mut ms_challenger, digests: *const u8, n: usize) -> i32;
    pub fn ms_challenger_sample_ext(ch: *mut ms_challenger, out2: *mut u64) -> i32;
    pub fn ms_challenger_sample_bits(ch: *mut ms_challenger, bits: u32, out: *mut u64) -> i32;
    pub fn ms_pcs_commit(ctx: *mut ms_ctx, log_blowup: u32, cap_height: u32, n: usize, evals: *const *const u64, heights: *const u64,
                         widths: *const u64, cap_out: *mut u8, out: *mut *mut ms_mmcs) -> i32;
    pub fn ms_pcs_open(ctx: *mut ms_ctx, params7: *const u64, n_rounds: usize, rounds: *const *mut ms_mmcs, n_points: *const u64,
                       points: *const u64, ch: *mut ms_challenger, opened_out: *mut u64, opened_cap_words: usize, fri_out: *mut u8,
                       fri_cap: usize, fri_len: *mut usize) -> i32;
    pub fn ms_pcs_verify(params7: *const u64, n_rounds: usize, caps: *const *const u8, cap_sizes: *const u64, n_mats: *const u64,
                         log_n: *const u64, widths: *const u64, n_points: *const u64, points: *const u64, opened: *const u64,
                         fri: *const u8, fri_len: usize, ch: *mut ms_challenger, accepted: *mut i32) -> i32;
    pub fn ms_trace_destroy(t: *mut ms_trace);
    pub fn ms_trace_info(t: *const ms_trace, out3: *mut u64) -> i32;
    pub fn ms_system_preprocessed_mmcs(sys: *mut ms_system, out: *mut *mut ms_mmcs) -> i32;
    pub fn ms_witness_commit_stage1(w: *mut ms_witness, cap_out: *mut u8, out: *mut *mut ms_mmcs) -> i32;
    pub fn ms_challenger_observe_claims(ch: *mut ms_challenger, w: *mut ms_witness) -> i32;
    pub fn ms_witness_claims_accumulator(w: *mut ms_witness, beta: *const u64, gamma: *const u64, acc_out: *mut u64) -> i32;
    pub fn ms_stage2_build(w: *mut ms_witness, beta: *const u64, gamma: *const u64, acc_in: *const u64, accs_out: *mut u64,
                           traces_out: *mut *mut ms_trace) -> i32;
    pub fn ms_pcs_commit_traces(ctx: *mut ms_ctx, log_blowup: u32, cap_height: u32, n: usize, evals: *const *mut ms_trace,
                                cap_out: *mut u8, out: *mut *mut ms_mmcs) -> i32;
    pub fn ms_quotient(sys: *mut ms_system, circuit: usize, log_n: u32, s1: *mut ms_mmcs, s1_idx: usize, s2: *mut ms_mmcs,
                       s2_idx: usize, publics8: *const u64, alpha: *const u64, q_lde_out: *mut *mut ms_trace) -> i32;
    pub fn ms_pcs_commit_ldes(ctx: *mut ms_ctx, cap_height: u32, n: usize, ldes: *const *mut ms_trace, cap_out: *mut u8,
                              out: *mut *mut ms_mmcs) -> i32;
    pub fn ms_stage2_trace(ctx: *mut ms_ctx, height: usize, num_lookups: usize, mult: *const u64, arg_offsets: *const u64,
                           args: *const u64, beta: *const u64, gamma: *const u64, acc_in: *const u64, trace_out: *mut u64,
                           acc_out: *mut u64) -> i32;
    pub fn ms_claims_accumulator(ctx: *mut ms_ctx, n_claims: usize, claim_offsets: *const u64, claim_data: *const u64, beta: *const u64,
                                 gamma: *const u64, acc_out: *mut u64) -> i32;
    pub fn ms_quotient_values(sys: *mut ms_system, circuit: usize, publics8: *const u64, log_n: u32, log_q: u32, pre_q: *const u64,
                              s1_q: *const u64, s2_q: *const u64, alpha: *const u64, out: *mut u64) -> i32;
    pub fn ms_field_op(ctx: *mut ms_ctx, op: i32, a: *const u64, b: *const u64, n: usize, out: *mut u64) -> i32;

    // ---- include/mstark_bb.h (BabyBear / degree-4 extension / Poseidon2: src/test_circuits/baby_bear_config.rs)
    pub fn msbb_system_create(ctx: *mut ms_ctx, blob: *const u8, len: usize, out: *mut *mut msbb_system) -> i32;
    pub fn msbb_system_destroy(sys: *mut msbb_system);
    pub fn msbb_system_preprocessed_commit(sys: *const msbb_system, out: *mut u32, cap_words: usize, n_digests: *mut usize) -> i32;
    pub fn msbb_system_circuit_info(sys: *const msbb_system, circuit: usize, out9: *mut u64) -> i32;
    pub fn msbb_witness_create(sys: *mut msbb_system, traces: *const *const u32, heights: *const u64, n_claims: usize,
                               claim_offsets: *const u64, claim_data: *const u32, out: *mut *mut msbb_witness) -> i32;
    pub fn msbb_witness_create_host(sys: *mut msbb_system, traces: *const *const u32, heights: *const u64, n_claims: usize,
                                    claim_offsets: *const u64, claim_data: *const u32, pinned: *mut i32, out: *mut *mut msbb_witness) -> i32;
    pub fn msbb_witness_destroy(w: *mut msbb_witness);
    pub fn msbb_prove(sys: *mut msbb_system, w: *mut msbb_witness, proof_out: *mut u8, cap: usize, proof_len: *mut usize,
                      stage_ms: *mut f64) -> i32;
    pub fn msbb_verify(sys: *mut msbb_system, n_claims: usize, claim_offsets: *const u64, claim_data: *const u32, proof: *const u8,
                       proof_len: usize, verdict: *mut i32) -> i32;
    pub fn msbb_set_poseidon2(ctx: *mut ms_ctx, constants141: *const u32) -> i32;
    pub fn msbb_poseidon2_permute(ctx: *mut ms_ctx, states: *mut u32, n: usize) -> i32;
    pub fn msbb_dft_batch(ctx: *mut ms_ctx, input: *const u32, h: usize, w: usize, inverse: i32, out: *mut u32) -> i32;
    pub fn msbb_coset_lde_batch(ctx: *mut ms_ctx, input: *const u32, h: usize, w: usize, log_blowup: u32, out: *mut u32) -> i32;
    pub fn msbb_mmcs_commit(ctx: *mut ms_ctx, n: usize, mats: *const *const u32, heights: *const u64, widths: *const u64,
                            cap_height: u32, cap_out: *mut u32, out: *mut *mut msbb_mmcs) -> i32;
    pub fn msbb_mmcs_open(m: *mut msbb_mmcs, index: usize, vals_out: *mut u32, proof_out: *mut u32, n_siblings: *mut usize) -> i32;
    pub fn msbb_mmcs_destroy(m: *mut msbb_mmcs);
    pub fn msbb_field_op(ctx: *mut ms_ctx, op: i32, a: *const u32, b: *const u32, n: usize, out: *mut u32) -> i32;
    // Level 2 of the BabyBear configuration (include/mstark_bb.h)
    pub fn msbb_challenger_create(sys: *mut msbb_system, out: *mut *mut msbb_challenger) -> i32;
    pub fn msbb_challenger_destroy(ch: *mut msbb_challenger);
    pub fn msbb_challenger_observe(ch: *mut msbb_challenger, elems: *const u32, n: usize) -> i32;
    pub fn msbb_challenger_observe_digests(ch: *mut msbb_challenger, digests: *const u32, n: usize) -> i32;
    pub fn msbb_challenger_sample_ext(ch: *mut msbb_challenger, out4: *mut u32) -> i32;
    pub fn msbb_challenger_sample_bits(ch: *mut msbb_challenger, bits: u32, out: *mut u64) -> i32;
    pub fn msbb_challenger_observe_claims(ch: *mut msbb_challenger, w: *mut msbb_witness) -> i32;
    pub fn msbb_trace_destroy(t: *mut msbb_trace);
    pub fn msbb_trace_info(t: *const msbb_trace, out3: *mut u64) -> i32;
    pub fn msbb_system_preprocessed_mmcs(sys: *mut msbb_system, out: *mut *mut msbb_mmcs) -> i32;
    pub fn msbb_witness_commit_stage1(w: *mut msbb_witness, cap_out: *mut u32, out: *mut *mut msbb_mmcs) -> i32;
    pub fn msbb_witness_claims_accumulator(w: *mut msbb_witness, beta: *const u32, gamma: *const u32, acc_out: *mut u32) -> i32;
    pub fn msbb_stage2_build(w: *mut msbb_witness, beta: *const u32, gamma: *const u32, acc_in: *const u32, accs_out: *mut u32,
                             traces_out: *mut *mut msbb_trace) -> i32;
    pub fn msbb_pcs_commit_traces(sys: *mut msbb_system, n: usize, evals: *const *mut msbb_trace, cap_out: *mut u32, out: *mut *mut msbb_mmcs) -> i32;
    pub fn msbb_quotient(sys: *mut msbb_system, circuit: usize, log_n: u32, s1: *mut msbb_mmcs, s1_idx: usize, s2: *mut msbb_mmcs, s2_idx: usize,
                         publics16: *const u32, alpha: *const u32, q_lde_out: *mut *mut msbb_trace) -> i32;
    pub fn msbb_pcs_commit_ldes(sys: *mut msbb_system, n: usize, ldes: *const *mut msbb_trace, cap_out: *mut u32, out: *mut *mut msbb_mmcs) -> i32;
    pub fn msbb_pcs_open(sys: *mut msbb_system, n_rounds: usize, rounds: *const *mut msbb_mmcs, n_points: *const u64, points: *const u32,
                         ch: *mut msbb_challenger, opened_out: *mut u32, opened_cap_words: usize, fri_out: *mut u8, fri_cap: usize,
                         fri_len: *mut usize) -> i32;
}
