/* mstark.h — C ABI of the MI355X (gfx950) prover for multi-stark's GoldilocksBlake3Config hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, returns an int32 status (0 = ok, <0 = error,
 * text via ms_last_error()) and never unwinds across the boundary: the reference's panics/asserts
 * (src/prover.rs:323-326,522-525,637-641; src/system.rs:149,171-178,249-264) become error codes.
 *
 * Data conventions: Goldilocks elements are canonical u64 (little-endian on the wire); an Ext2
 * value is two consecutive u64 (c0, c1) — the layout `flatten_to_base` produces (src/prover.rs:417,495);
 * matrices cross the ABI row-major (what `RowMajorMatrix.values` holds); digests are 32 raw bytes.
 * The caller owns every host buffer for the duration of a call; the library owns device memory behind the
 * opaque handles below. A ms_ctx is bound to one HIP device and is not thread-safe. Handles may be destroyed in any
 * order: a system keeps its context alive, a witness its system, an mmcs its context (the memory is released when the
 * last dependent handle is destroyed).
 *
 * What each group replaces in /root/reference:
 *   ms_system_*    System::new + ProverKey                      src/system.rs:115-203 (preprocessed commit :190-195)
 *   ms_witness_*   SystemWitness / from_stage_1                 src/system.rs:225-328 (host-resident: the witness argument
 *                                                               of prove(), src/prover.rs:290-295)
 *   ms_prove       System::prove_multiple_claims                src/prover.rs:290-603
 *   ms_verify      System::verify_multiple_claims               src/verifier.rs:208-532
 *   ms_prove_sharded   the same proof computed by several GPUs  src/prover.rs:290-603 (commit/open calls :350,419,526,580)
 *   ms_comm_rccl_*     its transport on RCCL (the reference has no collectives: Cargo.toml has no MPI / NCCL crate)
 *   ms_comm_local_*    its transport between threads of one process (peer copies; the reference's own process model)
 *   ms_dft_batch   Radix2DitParallel::dft_batch                 src/prover.rs:650,716 (type fixed at :440)
 *   ms_coset_lde_batch  the LDE inside Pcs::commit              src/prover.rs:350,419; layout pinned by :975-999
 *   ms_quotient_lde     shifted_quotient_slices + lde_from_shifted_coefficients   src/prover.rs:631-717
 *   ms_mmcs_*      MerkleTreeMmcs commit / open_batch           src/types.rs:82-83,202-207; Pcs::commit_ldes src/prover.rs:526
 *   ms_stage2_trace     LookupValues::stage_2_traces            src/lookup.rs:472-555
 *   ms_claims_accumulator   the claims loop                     src/prover.rs:382-387
 *   ms_quotient_values      quotient_values(+_inner)            src/prover.rs:756-962 (sweep: src/eval.rs:67-106;
 *                                                               logUp: src/lookup.rs:152-256)
 *   ms_pcs_commit / ms_pcs_open / ms_pcs_verify / ms_challenger_*   Pcs::commit / open / verify with the transcript as a handle
 *                                                               src/prover.rs:350,419,580; examples/pcs_example.rs:64-121
 *   ms_blake3      Blake3 as used by the challenger             src/types.rs:28-29 (HashChallenger<u8,Blake3,32>)
 */
#ifndef MSTARK_H
#define MSTARK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ms_ctx ms_ctx;         /* one HIP device + stream + twiddle tables + memory pool */
typedef struct ms_system ms_system;   /* System<GoldilocksBlake3Config> + ProverKey (device-resident preprocessed LDE/tree) */
typedef struct ms_witness ms_witness; /* SystemWitness + claims, resident in HBM */
typedef struct ms_mmcs ms_mmcs;       /* ProverData of one commitment */

enum {
  MS_OK = 0,
  MS_ERR = -1,            /* generic failure, see ms_last_error() */
  MS_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime failure at context creation */
  MS_ERR_BUFFER = -3      /* caller-provided capacity too small; the needed size is reported */
};

const char* ms_last_error(void);

/* ---- context */
/* HIP devices visible to this process (0 when there is none or no driver); never an error. A host that spreads one proof
 * over N devices (ms_prove_sharded with ms_comm_local_*) checks this first instead of running on fewer. */
int32_t ms_device_count(void);
int32_t ms_ctx_create(int32_t device, ms_ctx** out);
void ms_ctx_destroy(ms_ctx* ctx);
int32_t ms_ctx_sync(ms_ctx* ctx);
/* Diagnostics: how often the host has waited for the context's stream since the context was created (every read-back that a
 * proof needs on the host before it can go on, ms_ctx_sync itself included). A proof's count is the difference around it:
 * two for ms_prove at the bench size (opened values, FRI), and the joint prover is held to that plus two. */
int32_t ms_ctx_sync_count(ms_ctx* ctx, uint64_t* out);
/* release pooled device memory back to the driver */
int32_t ms_ctx_trim(ms_ctx* ctx);
/* Per-kernel-class timing with HIP events on the library's stream. mask bit i enables class i. */
int32_t ms_ctx_set_profile_mask(ms_ctx* ctx, uint32_t mask);
int32_t ms_ctx_kernel_stats(ms_ctx* ctx, int32_t kernel_id, uint64_t* launches, double* ms, double* alg_bytes);
/* work items other than bytes recorded for a class (Poseidon2 permutations of the hash classes on the BabyBear path) */
int32_t ms_ctx_kernel_units(ms_ctx* ctx, int32_t kernel_id, double* units);
int32_t ms_ctx_reset_stats(ms_ctx* ctx);
/* Diagnostics: the nth device allocation from now fails (0 = off). Lets a test check that an error in the middle of a
 * proof leaves the library usable (queued read-backs dropped, pool intact). */
int32_t ms_ctx_debug_fail_alloc(ms_ctx* ctx, int32_t nth);
int32_t ms_kernel_count(void);
const char* ms_kernel_name(int32_t kernel_id);

/* ---- System::new (src/system.rs:115-203). `blob` is the front-end's system blob: params, then per circuit the
 * compiled node program (graph::ConstraintGraph fields, src/graph.rs:62-76), lookups and preprocessed trace.
 * FriParameters::max_log_arity (src/types.rs:189-190,215) may be 1 .. 6: a commit-phase round folds 2^a values per row, a as
 * large as max_log_arity allows without stepping over the next input's height or below the final height; a row of 2^a
 * extension values is one leaf (at most one BLAKE3 chunk, hence 6). Every config, example and bench of the reference
 * sets 1 (benches/multi_stark.rs:252, examples/simple_proof.rs:54, pcs_example.rs:39, ...), which is the path the
 * device-side transcript and the fused round kernels serve; wider rounds are driven from the host (one synchronisation
 * per round). ms_system_create, ms_pcs_open and ms_pcs_verify (and their msbb_ counterparts) refuse other values with
 * an error code. */
int32_t ms_system_create(ms_ctx* ctx, const uint8_t* blob, size_t len, ms_system** out);
void ms_system_destroy(ms_system* sys);
/* preprocessed commitment (System.preprocessed_commit): writes n_digests * 32 bytes; n_digests = 0 if none */
int32_t ms_system_preprocessed_commit(const ms_system* sys, uint8_t* out, size_t cap, size_t* n_digests);
/* [main_width, pre_width, pre_height, num_lookups, stage2_width, constraint_count, max_constraint_degree,
 *  quotient_degree, args_width] */
int32_t ms_system_circuit_info(const ms_system* sys, size_t circuit, uint64_t out9[9]);

/* ---- SystemWitness (src/system.rs:225-233) + claims. traces[i]: heights[i] x main_width_i row-major (height 0 =
 * inactive circuit). mult[i] / args[i]: the flat LookupValues storage (src/lookup.rs:392-405); pass mult = NULL
 * to have the library run SystemWitness::from_stage_1 (src/system.rs:244-328) on the host. Claims are given as
 * offsets (n_claims + 1) into claim_data. Everything is uploaded once; ms_prove does not consume it. */
int32_t ms_witness_create(ms_system* sys, const uint64_t* const* traces, const uint64_t* heights,
                          const uint64_t* const* mult, const uint64_t* const* args, size_t n_claims,
                          const uint64_t* claim_offsets, const uint64_t* claim_data, ms_witness** out);
/* A SystemWitness that STAYS in host memory, which is what the reference's prove() is handed (src/prover.rs:290-295;
 * criterion builds it in the setup closure, benches/multi_stark.rs:292-296). Nothing is uploaded here: the values are
 * validated and the caller's trace buffers are page-locked (hipHostRegister; *pinned = 1 when every range could be, else
 * the uploads of this witness go through a page-locked bounce buffer of the context - no asynchronous copy ever reads
 * pageable caller memory). The locks are counted per range process-wide: several witnesses may be made from the same
 * buffers, the range stays locked until the last of them is destroyed; a range the application has registered itself is
 * left alone. The trace buffers must stay valid and unchanged until ms_witness_destroy, which waits for the context's
 * streams before it gives the locks up. Every
 * ms_prove on such a witness moves traces and claims to HBM on a copy stream (claims travel while stage 1 is computed),
 * runs SystemWitness::from_stage_1 (src/system.rs:244-328) on the device and frees the device copies again, so its
 * wall time is the reference's timed region: witness in host memory at the start, proof bytes in host memory at the end.
 * Narrow upload: a trace of at least 4 MB whose values all fit 1 / 2 / 4 bytes (seen by the validation pass here) is not
 * sent as 64-bit words: every ms_prove narrows it on a pool of host threads (16 by default), checking the range again, uploads the narrowed chunks as they complete and widens them on the device - at the
 * bench size 15 MB instead of 117 MB cross PCIe, in 0.5 ms instead of 2.1. A value that no longer fits sends that proof
 * down the plain path. Environment: MSAMD_NO_PACK=1 (read here) never narrows; MSAMD_PACK_THREADS=n (0: off),
 * MSAMD_PACK_MIN_BYTES, MSAMD_PACK_CHUNKS, MSAMD_PACK_AFFINITY=1 (workers confined to the NUMA node of the trace) tune it. */
int32_t ms_witness_create_host(ms_system* sys, const uint64_t* const* traces, const uint64_t* heights, size_t n_claims,
                               const uint64_t* claim_offsets, const uint64_t* claim_data, int32_t* pinned /* nullable */,
                               ms_witness** out);
/* Proving in a stream (a prover service proves one witness after another): with on = 1 every ms_prove of this host-resident
 * witness also queues the upload for the NEXT ms_prove of it, into a second set of device buffers, behind its own; that
 * upload travels over PCIe while the current proof is computed, and the next ms_prove starts with its inputs in HBM. The
 * first proof after switching on still uploads by itself; the last prefetch is unused; on = 0 drops a pending one. (The
 * host buffers are fixed for the life of the witness anyway, see ms_witness_create_host.) bench.py reports this mode as
 * config.pipelined_ms_per_proof, never as the primary figure (whose every step starts with the witness in host memory). */
int32_t ms_witness_prefetch(ms_witness* w, int32_t on);
/* The bench workload's witness and claims generated in HBM for the system [ByteTable, U32Add]: build_witness +
 * build_claims of benches/multi_stark.rs:171-238 (two xorshift32 streams from a0, b0; the reference uses 0xdeadbeef,
 * 0xcafebabe) followed by from_stage_1 on the device. Nothing crosses PCIe. */
int32_t ms_witness_u32_add_bench(ms_system* sys, size_t num_adds, uint32_t a0, uint32_t b0, ms_witness** out);
void ms_witness_destroy(ms_witness* w);

/* ---- System::prove_multiple_claims (src/prover.rs:290-603). Writes Proof::to_bytes (src/prover.rs:241-248).
 * stage_ms (optional, 6 doubles): stage1_commit, lookup_construction, stage2_commit, quotient, fri_open, total —
 * the reference's span names (src/prover.rs:336-538). Returns MS_ERR_BUFFER with *proof_len = needed size if
 * cap is too small. */
int32_t ms_prove(ms_system* sys, ms_witness* w, uint8_t* proof_out, size_t cap, size_t* proof_len, double* stage_ms);

/* ---- System::verify_multiple_claims (src/verifier.rs:208-532; shape checks :536-695) on the bytes ms_prove wrote.
 * Returns MS_OK when the check ran; *verdict = 0 if the proof is accepted, else the reference's VerificationError
 * variant (src/verifier.rs:176-192): 2 InvalidOpeningArgument, 3 InvalidProofShape, 4 InvalidSystem,
 * 5 OodEvaluationMismatch, 6 UnbalancedChannel. Claims as in ms_witness_create. The claims part (transcript hash and
 * initial accumulator) runs on the device, the rest on the host. */
#define MS_VERDICT_ACCEPT 0
#define MS_VERDICT_INVALID_OPENING 2
#define MS_VERDICT_INVALID_SHAPE 3
#define MS_VERDICT_INVALID_SYSTEM 4
#define MS_VERDICT_OOD_MISMATCH 5
#define MS_VERDICT_UNBALANCED 6
int32_t ms_verify(ms_system* sys, size_t n_claims, const uint64_t* claim_offsets, const uint64_t* claim_data,
                  const uint8_t* proof, size_t proof_len, int32_t* verdict);

/* ---- One proof over several GPUs (one process per GPU; SURVEY §8e, BASELINE config 3). The reference has no such
 * mode: this is System::prove_multiple_claims (src/prover.rs:290-603) with the Pcs::commit / Pcs::open calls
 * (:350,419,526,580) spread over ranks, producing the same Proof bytes as ms_prove on one device.
 * Every rank creates the SAME system and a witness that holds: the traces of the circuits it computes (its own
 * "sharded" circuit and every replicated one; traces[i] = NULL with heights[i] > 0 marks a circuit computed elsewhere),
 * the heights of all circuits, and all claims. owners[i] = rank that computes circuit i, or -1 = replicated on every
 * rank (small tables). A rank may own any number of circuits (none included) of any shapes: the reference's multi-circuit
 * systems (one wide circuit beside tables of other widths and heights, src/test_circuits/blake3.rs:2215-2613) split as they
 * are. Limits: the number of ranks is a power of two; every committed LDE has at least as many rows as there are ranks
 * (any cap_height: a cap taller than log2(ranks) is gathered from inside the ranks' sub-trees). When there is exactly one sharded circuit per rank, all of one shape, the k-th owned by rank k
 * (BASELINE config 3) the row ranges travel by one symmetric all-to-all per column group; otherwise every sharded matrix is
 * handed out by its owner (ms_comm.scatter_cols_start, which the transport must then offer).
 * The library calls back for the two exchanges it needs; both take DEVICE pointers of this context's device, are
 * called with the context's stream idle, and must have completed when they return (0 = ok):
 *   all_to_all: send/recv hold `world` blocks of bytes_per_peer bytes; block k of send goes to rank k, block k of recv
 *               comes from rank k (row ranges of the LDE matrices before leaf hashing);
 *   all_gather: every rank contributes `bytes`, recv gets world * bytes ordered by rank (sub-tree roots, logUp totals,
 *               opened values, reduced openings, query openings).
 * With torch.distributed these are all_to_all_single / all_gather_into_tensor on RCCL (multi-stark_amd/sharded.py). */
typedef struct ms_comm {
  /* sizeof(ms_comm) as the HOST compiled it. Members are only ever appended: the library reads the first `size` bytes and
   * treats every member beyond them as NULL (not offered), so a host built against an older header stays valid. A size that
   * does not even cover all_gather is refused. */
  uint32_t size;
  int32_t rank, world;
  void* user;
  int32_t (*all_to_all)(void* user, const void* send_dev, void* recv_dev, size_t bytes_per_peer);
  int32_t (*all_gather)(void* user, const void* send_dev, void* recv_dev, size_t bytes);
  /* Optional pair (both NULL = not offered): a non-blocking all_to_all, so that the exchange of one group of LDE columns
   * runs while the next group is being transformed. `start` is called with the context's stream idle and returns at once;
   * chunk k of the exchange is bytes_per_peer bytes at send_dev + k * send_stride (to rank k) and at recv_dev +
   * k * recv_stride (from rank k). Buffers stay untouched until `wait` - which completes every started exchange - returns. */
  int32_t (*all_to_all_start)(void* user, const void* send_dev, size_t send_stride, void* recv_dev, size_t recv_stride,
                              size_t bytes_per_peer);
  int32_t (*all_to_all_wait)(void* user);
  /* Optional (NULL = not offered; needs all_to_all_wait): the same non-blocking exchange read STRAIGHT OUT OF a column-major
   * matrix, so that no send buffer has to be packed first. For every rank k, `ncols` segments of seg_bytes bytes each way:
   * segment c for rank k is read at send_dev + k * send_peer_stride + c * send_col_stride (rows [k h / N, (k + 1) h / N)
   * of column c of this rank's LDE), and segment c coming from rank k is written at recv_dev + k * recv_peer_stride +
   * c * recv_col_stride (block k of this rank's receive buffer, column c). Segments of one peer travel in column order on
   * both sides; the rank's own segments (k = rank) are copied on the device. Completed by all_to_all_wait. */
  int32_t (*all_to_all_cols_start)(void* user, const void* send_dev, size_t send_peer_stride, size_t send_col_stride, void* recv_dev,
                                   size_t recv_peer_stride, size_t recv_col_stride, size_t ncols, size_t seg_bytes);
  /* Optional (NULL = not offered): order the transport with a HIP stream by events instead of by the host. After
   * set_stream_ordered(user, s) every call above (1) first makes the transport's own stream wait for what has been queued on
   * `s` so far - so the caller need not leave `s` idle - and (2) returns without waiting: the blocking calls and
   * all_to_all_wait make `s` wait for the transport's stream instead of the host. A joint proof synchronises with the host some
   * twenty times less often this way. set_stream_ordered(user, NULL) restores the blocking contract; ms_prove_sharded
   * switches the mode on for its own duration when the transport offers it (MSAMD_SHARDED_HOST_SYNC=1: never). */
  int32_t (*set_stream_ordered)(void* user, void* hip_stream);
  /* Optional (NULL = not offered): all_to_all_cols_start with flags. MS_COMM_SKIP_SELF: the rank's own segments (k = rank) are
   * NOT copied - the library hashes and reduces its own rows where they are, inside its LDE, so that block of the receive buffer
   * is never read (at world 1 the exchange then moves nothing at all). Without this member the library calls
   * all_to_all_cols_start; the self-copy is then made and ignored. */
  int32_t (*all_to_all_cols_start2)(void* user, const void* send_dev, size_t send_peer_stride, size_t send_col_stride, void* recv_dev,
                                    size_t recv_peer_stride, size_t recv_col_stride, size_t ncols, size_t seg_bytes, uint32_t flags);
  /* Optional (NULL = not offered; needs all_to_all_wait): ONE rank's matrix handed out by row ranges - what the library uses
   * when the sharded circuits differ in shape or in number per rank. Rank `root` sends, to every OTHER rank k, ncols segments
   * of seg_bytes read at send_dev + k * send_peer_stride + c * send_col_stride; every other rank receives its ncols
   * segments at recv_dev + c * recv_col_stride. Nothing is copied for k = root (its rows stay where they are); send_dev is
   * ignored on the other ranks, recv_dev on root. Non-blocking like all_to_all_start, completed by all_to_all_wait; every
   * rank makes the same sequence of calls. */
  int32_t (*scatter_cols_start)(void* user, int32_t root, const void* send_dev, size_t send_peer_stride, size_t send_col_stride,
                                void* recv_dev, size_t recv_col_stride, size_t ncols, size_t seg_bytes);
  /* Optional (NULL = not offered): this rank cannot go on - a validation that fails on ONE rank, an allocation failure, a
   * failed exchange - while its peers have entered, or are about to enter, the next exchange. ms_prove_sharded calls it on
   * every error path before it returns, so that the peers' pending and future calls FAIL instead of waiting for operations
   * that will never be issued (RCCL: ncclCommAbort; threads: ms_comm_local_group_abort). The transport is unusable
   * afterwards. Without it only the host's watchdog ends a proof one rank has left. Must not fail and must not throw. */
  void (*abort)(void* user, const char* why);
} ms_comm;
#define MS_COMM_SKIP_SELF 1u
int32_t ms_prove_sharded(ms_system* sys, ms_witness* w, const ms_comm* comm, const int32_t* owners, uint8_t* proof_out, size_t cap,
                         size_t* proof_len, double* stage_ms);
/* A rank of a joint proof need not hold ALL claims on the host (8.4 M claims on each of eight ranks at BASELINE config 3): it
 * reads the element range ms_claims_slice_range reports - the claims whose transcript words fall into its range of BLAKE3
 * chunks, cut from the system's shape, the circuits' heights and the n_claims + 1 element offsets (which every rank keeps
 * in full: 8 bytes per claim) - plus the first 130 elements, which every rank needs for the transcript's first chunk.
 * ms_witness_create_host_sliced is ms_witness_create_host with that part of claim_data only: data_slice holds the elements
 * [data_first, data_first + data_count), head the first min(total, 130). A list of at most 8192 transcript words is absorbed
 * whole by every rank (the range is then everything). Only ms_prove_sharded accepts such a witness. */
int32_t ms_claims_slice_range(ms_system* sys, const uint64_t* heights, size_t n_claims, const uint64_t* claim_offsets, int32_t rank,
                              int32_t world, uint64_t* first_elem, uint64_t* n_elems);
int32_t ms_witness_create_host_sliced(ms_system* sys, const uint64_t* const* traces, const uint64_t* heights, size_t n_claims,
                                      const uint64_t* claim_offsets, uint64_t data_first, uint64_t data_count,
                                      const uint64_t* data_slice, const uint64_t* head, size_t n_head, int32_t* pinned,
                                      ms_witness** out);

/* Where a joint proof is: the library records every call it makes into the transport - the prover's phase, the collective,
 * its size, the rank ("stage-2 commit: all_to_all_cols_start (29360128 bytes, part 3) on rank 5 of 8"). `out` receives the
 * text of the LAST call entered through this context, *seq how many have been entered since the context was created and
 * *in_flight whether that call has not returned yet. Callable from any thread while ms_prove_sharded runs on another (it is
 * what a watchdog prints when a peer never joins a collective: which exchange, on which rank). */
int32_t ms_ctx_comm_progress(ms_ctx* ctx, char* out, size_t cap, uint64_t* seq, int32_t* in_flight);

/* ---- Native transport for ms_prove_sharded: RCCL over xGMI (csrc/comm_rccl.hip; librccl is loaded at run time). The host
 * needs no Python: rank 0 draws the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by whatever channel the
 * host already has (a file, MPI, a TCP store); every rank then creates its transport on its own context (collective:
 * ncclCommInitRank) and passes ms_comm_rccl_table() to ms_prove_sharded. all_to_all = one grouped ncclSend / ncclRecv
 * pair per peer, all_gather = ncclAllGather, both on the transport's own stream; world = 1 needs no id and no librccl. */
#define MS_RCCL_UNIQUE_ID_BYTES 128
typedef struct ms_comm_rccl ms_comm_rccl;
int32_t ms_comm_rccl_unique_id(uint8_t out[MS_RCCL_UNIQUE_ID_BYTES]);
int32_t ms_comm_rccl_create(ms_ctx* ctx, const uint8_t unique_id[MS_RCCL_UNIQUE_ID_BYTES], int32_t rank, int32_t world,
                            ms_comm_rccl** out);
const ms_comm* ms_comm_rccl_table(ms_comm_rccl* c);
uint64_t ms_comm_rccl_bytes_moved(ms_comm_rccl* c); /* bytes this rank has put through the two exchanges so far */
void ms_comm_rccl_destroy(ms_comm_rccl* c);

/* ---- In-process transport for ms_prove_sharded (csrc/comm_local.hip): the ranks are THREADS of one process, each with its
 * own ms_ctx (one device each, or several contexts on one device), and the exchanges are device-to-device copies pulled by the
 * receiving rank, ordered between the ranks' streams by HIP events - peer copies over xGMI when the contexts sit on
 * different GPUs. For a host that drives a whole node from one process (the reference's prover is one process with a thread
 * pool, Cargo.toml:45) and for tests: RCCL refuses two ranks on one device, thread ranks do not, so BASELINE config 3 runs at
 * world 8 on a one-GPU box. Create the group once, then one handle per rank (any thread); every rank's thread passes
 * ms_comm_local_table() to ms_prove_sharded. A rank that fails or never arrives does not hang the others: the rendezvous times
 * out (MSAMD_LOCAL_TIMEOUT_S, default 120) and ms_comm_local_group_abort wakes every waiting rank with an error. Handles and
 * group may be destroyed in any order. */
typedef struct ms_comm_local_group ms_comm_local_group;
typedef struct ms_comm_local ms_comm_local;
int32_t ms_comm_local_group_create(int32_t world, ms_comm_local_group** out);
void ms_comm_local_group_abort(ms_comm_local_group* g);
void ms_comm_local_group_destroy(ms_comm_local_group* g);
int32_t ms_comm_local_create(ms_comm_local_group* g, ms_ctx* ctx, int32_t rank, ms_comm_local** out);
const ms_comm* ms_comm_local_table(ms_comm_local* c);
uint64_t ms_comm_local_bytes_moved(ms_comm_local* c);
void ms_comm_local_destroy(ms_comm_local* c);

/* ---- PCS-level entry points (host buffers in, host buffers out) */
/* out[k] = sum_j in[j] w_h^{jk} per column (inverse != 0: the inverse transform incl. 1/h); natural order both sides */
int32_t ms_dft_batch(ms_ctx* ctx, const uint64_t* in, size_t h, size_t w, int32_t inverse, uint64_t* out);
/* coset_lde_batch(evals, log_blowup, GENERATOR).bit_reverse_rows(): (h << log_blowup) x w */
int32_t ms_coset_lde_batch(ms_ctx* ctx, const uint64_t* in, size_t h, size_t w, uint32_t log_blowup, uint64_t* out);
/* quotient evaluations in natural order on the quotient domain ((n q) x D) -> committed LDE ((n B) x (q D)) */
int32_t ms_quotient_lde(ms_ctx* ctx, const uint64_t* in, uint32_t log_n, uint32_t log_q, uint32_t log_blowup, size_t D,
                        uint64_t* out);
/* Merkle commit over matrices of power-of-two heights (mixed heights allowed); cap_out: 32 << cap_height bytes */
int32_t ms_mmcs_commit(ms_ctx* ctx, size_t n, const uint64_t* const* mats, const uint64_t* heights,
                       const uint64_t* widths, uint32_t cap_height, uint8_t* cap_out, ms_mmcs** out);
/* open_batch(index): opened rows concatenated in matrix order; siblings bottom-up; *n_siblings written */
int32_t ms_mmcs_open(ms_mmcs* m, size_t index, uint64_t* vals_out, uint8_t* proof_out, size_t* n_siblings);
void ms_mmcs_destroy(ms_mmcs* m);
int32_t ms_blake3(ms_ctx* ctx, const uint8_t* bytes, size_t len, uint8_t out32[32]);

/* ---- Pcs::commit / Pcs::open / Pcs::verify on their own (examples/pcs_example.rs:64-121; the calls of src/prover.rs:350,419,580
 * for a host that keeps its own prover loop). The transcript is a handle: ms_challenger_create(params7) is
 * config.initialise_challenger() (src/types.rs:118-130) for the seven parameter words [log_blowup, cap_height,
 * log_final_poly_len, max_log_arity, num_queries, commit_proof_of_work_bits, query_proof_of_work_bits].
 * ms_pcs_commit: evaluations over the natural domains (row-major, heights[i] x widths[i]) -> coset LDE + Merkle tree, all
 * kept on the device behind ms_mmcs. ms_pcs_open: one ms_mmcs per round; n_points has one entry per matrix (rounds
 * flattened, at most two points per matrix), points holds (c0, c1) per point; opened values are written in
 * round -> matrix -> point -> column order (2 words each); fri_out receives the FriProof (the opening_proof field of
 * Proof::to_bytes); MS_ERR_BUFFER with *fri_len = needed size if a capacity is too small (the challenger has then been
 * advanced: restart from a fresh one). ms_pcs_verify: the same rounds described by their caps (32-byte digests), log2 domain
 * sizes and widths; *accepted = 1 / 0. */
typedef struct ms_challenger ms_challenger;
int32_t ms_challenger_create(const uint64_t params7[7], ms_challenger** out);
void ms_challenger_destroy(ms_challenger* ch);
int32_t ms_challenger_observe(ms_challenger* ch, const uint64_t* elems, size_t n);
int32_t ms_challenger_observe_digests(ms_challenger* ch, const uint8_t* digests, size_t n);
int32_t ms_challenger_sample_ext(ms_challenger* ch, uint64_t out2[2]);
int32_t ms_challenger_sample_bits(ms_challenger* ch, uint32_t bits, uint64_t* out);
int32_t ms_pcs_commit(ms_ctx* ctx, uint32_t log_blowup, uint32_t cap_height, size_t n, const uint64_t* const* evals,
                      const uint64_t* heights, const uint64_t* widths, uint8_t* cap_out, ms_mmcs** out);
int32_t ms_pcs_open(ms_ctx* ctx, const uint64_t params7[7], size_t n_rounds, ms_mmcs* const* rounds, const uint64_t* n_points,
                    const uint64_t* points, ms_challenger* ch, uint64_t* opened_out, size_t opened_cap_words, uint8_t* fri_out,
                    size_t fri_cap, size_t* fri_len);
int32_t ms_pcs_verify(const uint64_t params7[7], size_t n_rounds, const uint8_t* const* caps, const uint64_t* cap_sizes,
                      const uint64_t* n_mats, const uint64_t* log_n, const uint64_t* widths, const uint64_t* n_points,
                      const uint64_t* points, const uint64_t* opened, const uint8_t* fri, size_t fri_len, ms_challenger* ch,
                      int32_t* accepted);

/* ---- Level 2: the prover's steps on DEVICE HANDLES, for a host that keeps the reference's own prover loop
 * (src/prover.rs:290-603: "Rust host keeps the multi-circuit / lookup bookkeeping and calls kernels through FFI") and
 * calls the device once per step. Between the calls only commitments, accumulators and challenges cross the boundary;
 * traces, LDEs and trees stay in HBM behind ms_witness / ms_trace / ms_mmcs. Together with ms_challenger_* and ms_pcs_open
 * these are all the device calls of prove_multiple_claims (tests/test_gpu_level2.py drives exactly that loop from Python
 * and obtains the bytes ms_prove writes):
 *   ms_witness_commit_stage1      pcs.commit(stage-1 traces)                             src/prover.rs:338-350
 *   ms_challenger_observe_claims  the claims absorbed by the transcript                  src/prover.rs:369-373
 *   ms_witness_claims_accumulator the initial accumulator                                src/prover.rs:382-387
 *   ms_stage2_build               LookupValues::stage_2_traces -> evaluation handles     src/prover.rs:400, src/lookup.rs:472-555
 *   ms_pcs_commit_traces          pcs.commit(stage-2 traces) on those handles            src/prover.rs:414-419
 *   ms_quotient                   quotient_values + shifted_quotient_slices +            src/prover.rs:459-468,483,511-517
 *                                 lde_from_shifted_coefficients -> LDE handle
 *   ms_pcs_commit_ldes            pcs.commit_ldes                                        src/prover.rs:526
 *   ms_system_preprocessed_mmcs   the ProverKey's preprocessed prover data (a view)      src/system.rs:190-195
 * Matrices of a commitment are indexed by ACTIVE position (circuits with an empty trace are skipped), as the reference's
 * per-stage matrix lists are (src/prover.rs:216-225). A handle consumed by a commit call keeps existing but is empty;
 * destroy it like any other. Host-resident witnesses (ms_witness_create_host) are not accepted here. */
typedef struct ms_trace ms_trace;
void ms_trace_destroy(ms_trace* t);
int32_t ms_trace_info(const ms_trace* t, uint64_t out3[3]); /* height, width, kind (0 evaluations, 1 LDE) */
int32_t ms_system_preprocessed_mmcs(ms_system* sys, ms_mmcs** out); /* *out = NULL when the system has no preprocessed trace */
int32_t ms_witness_commit_stage1(ms_witness* w, uint8_t* cap_out, ms_mmcs** out);
int32_t ms_challenger_observe_claims(ms_challenger* ch, ms_witness* w);
int32_t ms_witness_claims_accumulator(ms_witness* w, const uint64_t beta[2], const uint64_t gamma[2], uint64_t acc_out[2]);
/* accs_out: 2 words per active circuit (the accumulator after each circuit, src/lookup.rs:544-550); traces_out: one handle
 * per active circuit */
int32_t ms_stage2_build(ms_witness* w, const uint64_t beta[2], const uint64_t gamma[2], const uint64_t acc_in[2], uint64_t* accs_out,
                        ms_trace** traces_out);
int32_t ms_pcs_commit_traces(ms_ctx* ctx, uint32_t log_blowup, uint32_t cap_height, size_t n, ms_trace* const* evals, uint8_t* cap_out,
                             ms_mmcs** out);
/* publics8 = [beta, gamma, acc_in, acc_out] as (c0, c1) pairs (src/lookup.rs:78-84); s1 / s2: the stage commitments and
 * the circuit's matrix index inside each */
int32_t ms_quotient(ms_system* sys, size_t circuit, uint32_t log_n, ms_mmcs* s1, size_t s1_idx, ms_mmcs* s2, size_t s2_idx,
                    const uint64_t publics8[8], const uint64_t alpha[2], ms_trace** q_lde_out);
int32_t ms_pcs_commit_ldes(ms_ctx* ctx, uint32_t cap_height, size_t n, ms_trace* const* ldes, uint8_t* cap_out, ms_mmcs** out);

/* ---- in-tree kernels of the reference, host buffers in and out (kernel-level parity tests) */
int32_t ms_stage2_trace(ms_ctx* ctx, size_t height, size_t num_lookups, const uint64_t* mult,
                        const uint64_t* arg_offsets, const uint64_t* args, const uint64_t beta[2],
                        const uint64_t gamma[2], const uint64_t acc_in[2], uint64_t* trace_out, uint64_t acc_out[2]);
int32_t ms_claims_accumulator(ms_ctx* ctx, size_t n_claims, const uint64_t* claim_offsets, const uint64_t* claim_data,
                              const uint64_t beta[2], const uint64_t gamma[2], uint64_t acc_out[2]);
/* traces given as natural-order evaluations on the quotient domain ((n q) rows each, row-major); out: (n q) x 2 */
int32_t ms_quotient_values(ms_system* sys, size_t circuit, const uint64_t publics8[8], uint32_t log_n, uint32_t log_q,
                           const uint64_t* pre_q, const uint64_t* s1_q, const uint64_t* s2_q, const uint64_t alpha[2],
                           uint64_t* out);
/* element-wise field ops on device, for arithmetic known-answer tests: op 0 add, 1 sub, 2 mul, 3 inverse(a),
 * 4 ext2 mul (pairs), 5 ext2 inverse (pairs) */
int32_t ms_field_op(ms_ctx* ctx, int32_t op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* MSTARK_H */
