#!/usr/bin/env python3
"""bench.py — prove() throughput of the U32-add + byte-table workload (benches/multi_stark.rs, bench_config())
on MI355X.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N = 1 (BASELINE config 2): one step = one System::prove_multiple_claims over the reference's timed region (SURVEY §8d,
benches/multi_stark.rs:292-296): the witness (traces + claims) is in pinned HOST memory when the step starts, the proof
bytes are in host memory when it ends; upload, from_stage_1 on the device and read-back are inside. The HBM-resident
figure (witness uploaded once, outside) is reported beside it as config.hbm_resident_ms.

N > 1 (BASELINE config 3, weak scaling): one step = ONE proof of the system [ByteTable, U32Add x N] computed by all
ranks together (ms_prove_sharded): rank k computes adder k (2^20 rows, its own xorshift seeds), every commitment is
one Merkle tree over all matrices, so the LDE row ranges are exchanged before leaf hashing (all-to-all), sub-tree
roots / logUp totals / opened values / reduced openings are all-gathered, and every rank returns the same proof bytes.
The exchanges run on the library's own RCCL transport (csrc/comm_rccl.hip: grouped ncclSend / ncclRecv and ncclAllGather
called from C); torch.distributed only bootstraps (the 128-byte RCCL id, the barrier, the max over ranks). As at N = 1 the
witness is host-resident: every step uploads the rank's own trace and its 1/N slice of the claims. The independent-proof mode
(one [ByteTable, U32Add] proof per GPU, host-resident witnesses, commitments all-gathered) is measured in the same run
and reported as the secondary object `replicas`. Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

# the pool's host driver supports dmabuf IPC only: RCCL (and any sharing of device memory across processes) needs this before
# the first HIP call of the process; exported on the boxes already, set here for a launcher that drops the environment
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, load_oracle  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
ALG_BYTES_PER_ROW = 5512
TRAFFIC_FILES = ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json")
WORKLOAD = ("U32-add + byte-table lookup (benches/multi_stark.rs), 2^%d additions per %s, bench_config(): log_blowup 2, "
            "100 queries, 10+10 PoW bits, GoldilocksBlake3Config; %s, proof bytes returned to host")
HOST_RESIDENT = ("witness (traces + claims, 64-bit words) in pinned host memory at step start: upload, from_stage_1 on the device "
                 "and read-back inside the timed region; the library narrows byte-valued traces on host threads INSIDE the step "
                 "before they cross PCIe (config.upload)")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="goldilocks", choices=["goldilocks", "babybear"], help="goldilocks: BASELINE config 2 / 3 (the "
                    "headline); babybear: BASELINE config 4, the reference's second StarkGenericConfig (BabyBear, degree-4 extension, "
                    "Poseidon2; src/test_circuits/baby_bear_config.rs) on MulAir at 2^log-adds rows, N = 1 only")
    ap.add_argument("--log-adds", type=int, default=20, help="log2 of U32 additions per proof / per rank (BASELINE: 20)")
    ap.add_argument("--cpu-log-adds", type=int, default=20, help="size of the CPU baseline leg (same workload as the GPU by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-in-flight", action="store_true", help="N = 1: skip the secondary two-proofs-in-flight throughput figure")
    ap.add_argument("--no-config4", action="store_true", help="N = 1: skip the secondary BASELINE-config-4 (BabyBear / Poseidon2) leg of the default run")
    ap.add_argument("--hbm-resident", action="store_true", help="N = 1: primary figure from a witness already resident in HBM "
                    "(round-1 definition) instead of the host-resident one")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL on ROCm; gloo only to "
                    "rehearse the multi-rank path with several ranks sharing one GPU: the exchanges then go through TorchComm)")
    ap.add_argument("--transport", default=None, choices=["local", "rccl", "torch"], help="N > 1: who runs the joint proof's exchanges. "
                    "local (default without a launcher): N thread ranks of THIS process, one device each, on the library's in-process "
                    "transport (ms_comm_local_*: peer copies ordered by HIP events) - the reference is one process driving the box "
                    "(Cargo.toml:45); rccl (default under torch.distributed.run): the library's own RCCL transport, one process per GPU; "
                    "torch: torch.distributed callbacks")
    ap.add_argument("--share-devices", action="store_true", help="N > 1, --transport local: REHEARSAL only - place the N thread ranks on "
                    "the devices there are (round-robin) instead of refusing to run on fewer than N; the line says so (config.ranks)")
    ap.add_argument("--joint", action="store_true", help="(default for N > 1) the joint proof is the primary figure")
    ap.add_argument("--replicas-primary", action="store_true", help="N > 1: one independent proof per rank as the primary figure")
    ap.add_argument("--no-replicas-leg", action="store_true", help="N > 1: skip the secondary independent-proof measurement")
    ap.add_argument("--no-joint-leg", action="store_true", help="N > 1 with --replicas-primary: skip the joint proof")
    ap.add_argument("--primary-timeout", type=float, default=600.0, help="N > 1: seconds after which a leg that cannot finish (a "
                    "failed rank leaves the others in a collective) ends the job with a non-zero status")
    return ap.parse_args()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class Watchdog:
    """Ends the whole job with a NON-ZERO status when a leg cannot finish: one rank failing inside a collective would leave
    the others blocked forever, and a hang must not be recorded as a success. Whatever rank 0 has measured so far is
    printed first, with the error recorded."""

    def __init__(self, rank, what, seconds, partial, ctx=None, soft=False):
        self.done = threading.Event()
        self.rank, self.what, self.seconds, self.partial, self.ctx = rank, what, seconds, partial, ctx
        self.soft = soft  # a SECONDARY leg: what was measured before it is complete - print it and end with status 0
        self.leg = "setup"
        threading.Thread(target=self._run, daemon=True).start()

    def where(self):
        """which collective this rank is in (the library records every call into the transport: ms_ctx_comm_progress)"""
        if self.ctx is None:
            return "leg '%s'" % self.leg
        try:
            text, seq, inside = self.ctx.comm_progress()
        except Exception as e:  # noqa: BLE001
            return "leg '%s' (no progress record: %s)" % (self.leg, e)
        if not seq:
            return "leg '%s', no exchange of the joint prover entered yet" % self.leg
        return "leg '%s', %s exchange #%d of this process: %s" % (self.leg, "INSIDE" if inside else "after", seq, text)

    def _run(self):
        if self.done.wait(self.seconds):
            return
        where = self.where()
        log("[rank %d] %s did not finish within %.0f s (%s): exiting with status %d" % (self.rank, self.what, self.seconds, where, 0 if self.soft else 3))
        if self.rank == 0:
            line = self.partial()
            if line is not None:
                why = "%s timed out after %.0f s (collective hang or a failed rank); rank 0 was in %s" % (self.what, self.seconds, where)
                if self.soft:
                    line["secondary_leg_error"] = why  # the primary figures above it stand
                else:
                    line["error"] = why
                print(json.dumps(line), flush=True)
        os._exit(0 if self.soft else 3)

    def finish(self):
        self.done.set()


# Classes of the Goldilocks path that gather SEVERAL kernels (compress_layer: compress3_k, three sub-tree kernels and the fused FRI
# rounds; stage2: terms, scans, write; other): their totals are logged, but the roofline line is about the dominant KERNEL, and a
# class of four kernels that edges past the largest single one by a few per cent from run to run (0.95 against 0.92 ms per proof)
# is not that. The BabyBear leg keeps every class (its compress_layer is one permutation kernel in two launch shapes).
MULTI_KERNEL_CLASSES = ("compress_layer", "stage2", "other")


def profile_first_step(ctx, step, rank, single_kernel_only=True):
    """one untimed step with HIP events around every kernel class: picks the dominant kernel (class of one kernel)"""
    names = ctx.kernel_names()
    ctx.set_profile(names)
    ctx.reset_stats()
    proof = step()
    table = ctx.kernel_stats()
    ctx.set_profile([])
    ranked = sorted(table.items(), key=lambda kv: -kv[1]["ms"])
    eligible = [kv for kv in ranked if not (single_kernel_only and kv[0] in MULTI_KERNEL_CLASSES)]
    dominant = eligible[0][0] if eligible and eligible[0][1]["ms"] > 0 else "ntt12_dif"  # the kernel that really took the most time
    if rank == 0:
        log("per-kernel-class device time of one proof (HIP events, profiled warmup step):")
        for n, s in ranked:
            if s["launches"]:
                log("  %-16s launches %4d  total %8.3f ms  alg %.1f GB/s" % (n, s["launches"], s["ms"], s["alg_bytes"] / max(s["ms"], 1e-9) / 1e6))
    return proof, dominant


VALU_FILES = ("r04_valu.json", "r03_valu.json", "r02_valu.json")
VALU_PEAK_TOPS = 39.4  # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz: one integer VALU instruction per lane and clock


# algorithmic bytes per launch of the transform classes in the committed counter pass (config 2: 42 columns x 2^22 rows x 16 B
# in 3 launches per pass); files written by this round's tools carry the figure themselves (alg_bytes_per_launch)
PROFILE_ALG_BYTES = {"ntt12_dif": 939524096.0, "ntt8s_dif": 939524096.0}


def measured_valu(kernel, bytes_per_launch=None):
    """VALU lane-operations per launch of `kernel` from the committed SQ_INSTS_VALU pass (tools/valu_from_pmc.py). The pass
    is of config 2's launches; a launch of another size (the joint prover transforms its columns in four groups) is priced by
    the class's lane-operations per algorithmic byte, which does not depend on the launch size."""
    for name in VALU_FILES:
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))[kernel]
        except Exception:
            continue
        ops = rec["valu_lane_ops_per_launch"]
        ref = rec.get("alg_bytes_per_launch") or PROFILE_ALG_BYTES.get(kernel)
        if bytes_per_launch and ref:
            return ops * bytes_per_launch / ref, name
        return (ops, name) if not bytes_per_launch else (None, name)
    return None, None


def roofline_of(dominant, dom, full_size=True):
    avg_ms = dom["ms"] / max(dom["launches"], 1)
    bytes_per_launch = dom["alg_bytes"] / max(dom["launches"], 1)
    achieved = bytes_per_launch / max(avg_ms, 1e-12) / 1e6  # GB/s
    line = _roofline_hbm(dominant, dom, avg_ms, bytes_per_launch, achieved, full_size)
    ops, valu_file = measured_valu(dominant, None if full_size else bytes_per_launch)
    if ops:
        # the bound that binds: these kernels are integer arithmetic (no 64-bit multiplier on gfx950), the
        # vector ALU issues at its peak long before HBM is busy. Counted instructions (every instruction as ONE issue slot,
        # although a v_mad_u64_u32 takes about two) against the issue peak.
        tops = ops / max(avg_ms, 1e-12) / 1e9
        line["valu"] = {
            "what": "VALU lane-operations per launch (SQ_INSTS_VALU x 64 of the committed counter pass profiles/%s, same workload) "
                    "%s/ this run's average launch time, against the integer issue peak" % (
                        valu_file, "" if full_size else "scaled by algorithmic bytes to this run's launch size "),
            "lane_ops_per_launch": ops, "achieved_Tops": tops, "peak_Tops": VALU_PEAK_TOPS, "frac": tops / VALU_PEAK_TOPS,
        }
        clock_name = next((n for n in ("r04_clock_valu.txt", "r03_clock_valu.txt") if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
        if clock_name:
            line["valu"]["peak_note"] = ("peak_Tops is at the nominal 2.4 GHz; under these kernels the card holds 2.2-2.3 GHz, and the committed pass "
                                         "profiles/" + clock_name + " (GRBM_GUI_ACTIVE / 8 / duration, SQ_INSTS_VALU of the same dispatches) gives each "
                                         "class's issue rate against the peak at the clock it ran at")
    return line


def _roofline_hbm(dominant, dom, avg_ms, bytes_per_launch, achieved, full_size=True):
    traffic, tfile = measured_traffic(dominant)
    note = "rocprofv3 --pmc passes of this command, committed; not re-measured in this run"
    ref = PROFILE_ALG_BYTES.get(dominant)
    if traffic is not None and not full_size:
        # the committed pass is of config 2's launches: another launch size is priced by the measured traffic per algorithmic byte
        traffic, note = (traffic * bytes_per_launch / ref, note + "; scaled by algorithmic bytes to this run's launch size") if ref else (None, note)
    return {
        "kernel": dominant,
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": "profiles/%s (%s)" % (tfile, note),
        "avg_launch_ms": avg_ms,
        "alg_bytes_per_launch": bytes_per_launch,
        "launches": dom["launches"],
    }


def base_line(args, n_gpus, value, ms_per_step):
    return {
        "metric": "prove_trace_rows_per_sec",
        "value": value,
        "unit": "rows/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
    }


# ------------------------------------------------------------------------------------------------ N = 1
def single_gpu(args, pkg, fe, ctx, torch):
    num_adds = 1 << args.log_adds
    t = time.time()
    traces, claims = fe.u32_add_bench_witness(num_adds)
    system = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    packed = fe.pack_claims(claims)
    # setup, untimed (criterion's setup closure builds the witness): validate + page-lock the host buffers. Every timed step
    # uploads them and runs SystemWitness::from_stage_1 on the device.
    witness = system.witness(traces, packed) if args.hbm_resident else system.host_witness(traces, packed)
    if not args.hbm_resident and not witness.pinned:
        log("warning: host buffers could not be page-locked; uploads are staged by the runtime")
    rows = witness.rows
    log("witness ready in %.1fs: %d rows/proof" % (time.time() - t, rows))

    def step():
        return system.prove_multiple_claims(witness)

    def sync():
        ctx.sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    proof, dominant = profile_first_step(ctx, step, 0)
    for _ in range(max(args.warmup, 1) - 1):
        proof = step()
    log("warm-up done (%d steps); timing %d steps" % (max(args.warmup, 1), args.steps))
    # ---- timed region: exactly K steps, HIP events only around the dominant kernel class
    ctx.set_profile([dominant])
    ctx.reset_stats()
    sync()
    per = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        proof = step()  # (returns with the proof bytes in host memory: a step is complete when it returns)
        per.append(1e3 * (time.perf_counter() - t1))
    sync()
    elapsed = time.perf_counter() - t0
    log("timed region done: %.3f ms per step (min %.3f, median %.3f, max %.3f)" % (1e3 * elapsed / args.steps, min(per), float(np.median(per)), max(per)))
    dom = ctx.kernel_stats()[dominant]
    ctx.set_profile([])
    stage = system.prove_multiple_claims(witness, want_times=True).stage_ms
    verdict = system.verify_multiple_claims(packed, proof.to_bytes())  # untimed: System::verify_multiple_claims on the bytes
    assert verdict == 0, "the library's own verifier rejects the proof (code %d)" % verdict
    # the same proof from a witness that already sits in HBM (round-1 definition of the step): context, never `value`
    hbm_ms = None
    if not args.hbm_resident:
        dw = system.witness(traces, packed)
        assert system.prove_multiple_claims(dw).to_bytes() == proof.to_bytes()
        k = max(3, min(args.steps, 10))
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(k):
            system.prove_multiple_claims(dw)
        ctx.sync()
        hbm_ms = 1e3 * (time.perf_counter() - t1) / k
        del dw
        log("HBM-resident witness: %.3f ms per proof" % hbm_ms)
    # proving in a stream: each proof uploads the next one's inputs while it computes (ms_witness_prefetch); context only
    pipe_ms = None
    if not args.hbm_resident:
        witness.prefetch(True)
        assert step().to_bytes() == proof.to_bytes() and step().to_bytes() == proof.to_bytes()
        k = max(3, min(args.steps, 10))
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(k):
            step()
        ctx.sync()
        pipe_ms = 1e3 * (time.perf_counter() - t1) / k
        witness.prefetch(False)
        log("host-resident witness with the next upload overlapped: %.3f ms per proof" % pipe_ms)
    # the same step with the narrow upload switched off (every 64-bit word crosses PCIe): context only
    plain_ms = None
    if not args.hbm_resident:
        os.environ["MSAMD_NO_PACK"] = "1"
        try:
            pw = system.host_witness(traces, packed)
        finally:
            del os.environ["MSAMD_NO_PACK"]
        assert system.prove_multiple_claims(pw).to_bytes() == proof.to_bytes()
        k = max(3, min(args.steps, 10))
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(k):
            system.prove_multiple_claims(pw)
        ctx.sync()
        plain_ms = 1e3 * (time.perf_counter() - t1) / k
        del pw
        log("host-resident witness, plain 64-bit upload: %.3f ms per proof" % plain_ms)
    # a prover SERVICE keeps two proofs in flight on the GPU (a second host thread on a second ms_ctx of the same device): one
    # proof's host round trips and latency-bound FRI rounds are filled by the other one's transforms. Throughput only - the
    # latency of a proof doubles - so it is context beside `value` (which stays one proof at a time), never `value`.
    in_flight = None
    if not args.hbm_resident and not args.no_in_flight:
        try:
            in_flight = two_in_flight(pkg, fe, system, witness, traces, packed, proof.to_bytes(), rows, max(5, min(args.steps, 20)))
            log("two proofs in flight: %.3f ms per proof (%.1f M rows/s)" % (in_flight["ms_per_proof"], in_flight["rows_per_s"] / 1e6))
        except Exception as e:  # noqa: BLE001  (a secondary figure must not void the run, but its failure is part of the record)
            log("two-in-flight leg failed: %r" % (e,))
            in_flight = {"error": "two-in-flight leg failed: %r" % (e,)}
    trace_bytes = int(sum(t.nbytes for t in traces))
    narrow = [int(t.nbytes // 8 * (1 if int(t.max(initial=0)) < 256 else 2 if int(t.max(initial=0)) < 65536 else 4 if int(t.max(initial=0)) < (1 << 32) else 8))
              if t.nbytes >= (4 << 20) else int(t.nbytes) for t in traces]
    result = base_line(args, 1, rows * args.steps / elapsed, 1e3 * elapsed / args.steps)
    result["config"] = {
        "workload": WORKLOAD % (args.log_adds, "proof", "witness resident in HBM" if args.hbm_resident else HOST_RESIDENT),
        "rows_per_proof": rows,
        "proof_bytes": len(proof.to_bytes()),
        "verified": True,
        "parallelism": "single GPU",
        "stage_ms": {k: round(v, 3) for k, v in stage.items()},
        "step_ms_min_median_max": [round(min(per), 3), round(float(np.median(per)), 3), round(max(per), 3)],  # box noise: the narrowing runs on shared host cores
        "whole_path_alg_GBps": ALG_BYTES_PER_ROW * rows / (elapsed / args.steps) / 1e9,
        "hbm_resident_ms": hbm_ms,
        "pipelined_ms_per_proof": pipe_ms,
        "two_in_flight": in_flight,
        "host_witness_bytes_per_proof": None if args.hbm_resident else int(trace_bytes + packed[0].nbytes + packed[1].nbytes),
        "upload": None if args.hbm_resident else {
            "what": "traces of at least 4 MB whose values all fit 1 / 2 / 4 bytes are narrowed by %s host threads inside the timed "
                    "step (range-checked again every proof), uploaded chunk by chunk and widened on the device; claims cross as "
                    "64-bit words behind stage 1" % os.environ.get("MSAMD_PACK_THREADS", "16"),
            "bytes_over_pcie_per_proof": int(sum(narrow) + packed[0].nbytes + packed[1].nbytes),
            "plain_upload_ms_per_proof": plain_ms,
        },
    }
    result["roofline"] = roofline_of(dominant, dom, full_size=(args.log_adds == 20))  # (the committed counter pass is of 2^20 additions)
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(fe, system.blob, args.cpu_log_adds)
    return result


def two_in_flight(pkg, fe, system, witness, traces, packed, want, rows, k):
    """k host-resident proofs per thread on two contexts of the same device at once (tools/two_in_flight.py)"""
    ctx2 = pkg.Context(0)
    system2 = pkg.System.new(ctx2, fe.bench_params(), fe.u32_add_system_inputs())
    witness2 = system2.host_witness([t.copy() for t in traces], packed)
    assert system2.prove_multiple_claims(witness2).to_bytes() == want
    slots = [(system, witness), (system2, witness2)]
    barrier = threading.Barrier(3)
    errors = []

    def body(i):
        sysm, w = slots[i]
        try:
            for _ in range(2):
                sysm.prove_multiple_claims(w)
            barrier.wait()
            for _ in range(k):
                p = sysm.prove_multiple_claims(w)
            assert p.to_bytes() == want
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    ths = [threading.Thread(target=body, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    del witness2, system2
    if errors:
        raise errors[0]
    return {"what": "two proofs in flight on one GPU (two host threads, two ms_ctx of the same device, host-resident witnesses): "
                    "throughput of a prover service; the latency of each proof is about twice ms_per_proof",
            "proofs": 2 * k, "ms_per_proof": 1e3 * dt / (2 * k), "rows_per_s": rows * 2 * k / dt}


# ------------------------------------------------------------------------------------------------ config 4
VALU_PEAK_TOPS = 39.4      # integer lane-ops/s at one VALU instruction per cycle and SIMD lane (tools/micro/b3_rate.hip: 58 G BLAKE3/s x 680)
POSEIDON2_INSTR = 9.0e3    # VALU instructions per Poseidon2-16 permutation and lane (profiles/r01_config4_kernel_stats.txt: SQ counters)


def babybear(args, pkg, fe, ctx, torch):
    """BASELINE config 4: BabyBear + degree-4 extension + Poseidon2 (the reference's test-suite instantiation,
    src/test_circuits/baby_bear_config.rs:28-127), MulAir at 2^20 rows with the test parameters (blowup 2, 64 queries). As for
    config 2 the step is the reference's timed region: the witness (12.6 MB of canonical u32 words) is in pinned HOST memory
    when the step starts (msbb_witness_create_host) and is uploaded inside it; --hbm-resident keeps it on the device."""
    bb = pkg.babybear
    consts = fe.poseidon2_constants()
    rows = 1 << args.log_adds
    with fe.field(fe.BABYBEAR):
        params = fe.test_params()
        system = bb.System.new(ctx, params, fe.mul_air_inputs(), consts)
        trace = fe.mul_air_trace(rows)
        packed = fe.pack_claims([])
    witness = system.witness([trace], packed) if args.hbm_resident else system.host_witness([trace], packed)

    def step():
        return system.prove_multiple_claims(witness)

    proof, dominant = profile_first_step(ctx, step, 0, single_kernel_only=False)
    for _ in range(max(args.warmup, 1) - 1):
        proof = step()
    # ---- timed region: exactly K steps, NO HIP events (this configuration's dominant class is ~150 short launches per proof:
    # events around each of them cost about a millisecond per proof, which is not the prover's time)
    ctx.set_profile([])
    ctx.sync()
    per = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        proof = step()
        per.append(1e3 * (time.perf_counter() - t1))
    ctx.sync()
    elapsed = time.perf_counter() - t0
    log("timed region done: %.3f ms per step (min %.3f, median %.3f, max %.3f)" % (1e3 * elapsed / args.steps, min(per), float(np.median(per)), max(per)))
    # ---- the dominant class's launch times: a separate pass with events around that class only, outside the timed region
    prof_steps = max(2, min(args.steps, 5))
    ctx.set_profile([dominant])
    ctx.reset_stats()
    ctx.sync()
    t1 = time.perf_counter()
    for _ in range(prof_steps):
        step()
    ctx.sync()
    prof_ms = 1e3 * (time.perf_counter() - t1) / prof_steps
    dom = ctx.kernel_stats()[dominant]
    ctx.set_profile([])
    assert system.verify(packed, proof.to_bytes()) == 0, "the library's own verifier rejects the proof"
    stage = system.prove_multiple_claims(witness, want_times=True).stage_ms
    hbm_ms = None
    if not args.hbm_resident:  # the same proof from a witness already in HBM: context, never `value`
        dw = system.witness([trace], packed)
        assert system.prove_multiple_claims(dw).to_bytes() == proof.to_bytes()
        k = max(3, min(args.steps, 10))
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(k):
            system.prove_multiple_claims(dw)
        ctx.sync()
        hbm_ms = 1e3 * (time.perf_counter() - t1) / k
        del dw
    result = base_line(args, 1, rows * args.steps / elapsed, 1e3 * elapsed / args.steps)
    result["dtype"] = "u32"
    result["config"] = {
        "workload": "BASELINE config 4: MulAir (a * b = c with a self-cancelling lookup pair), 2^%d rows, BabyBear / degree-4 extension / "
                    "Poseidon2-16 sponge and compression / DuplexChallenger (src/test_circuits/baby_bear_config.rs), log_blowup 1, 64 queries; "
                    "%s, proof bytes returned to host" % (args.log_adds, "witness resident in HBM" if args.hbm_resident else
                                                           "witness (canonical u32 words) in pinned host memory at step start, uploaded inside the step"),
        "rows_per_proof": rows,
        "proof_bytes": len(proof.to_bytes()),
        "verified": True,
        "parallelism": "single GPU",
        "stage_ms": {k: round(v, 3) for k, v in stage.items()},
        "step_ms_min_median_max": [round(min(per), 3), round(float(np.median(per)), 3), round(max(per), 3)],
        "hbm_resident_ms": hbm_ms,
    }
    rl = roofline_of(dominant, dom, full_size=False)
    rl["traffic"], rl["traffic_source"] = None, "not collected for this configuration"
    rl["timed_how"] = ("HIP events around every launch of this class in a SEPARATE pass of %d steps after the timed region (%.3f ms per step "
                       "with the events, %.3f without): the class is %d launches per proof, and events around them inside the timed region "
                       "would add their own cost to ms_per_step" % (prof_steps, prof_ms, 1e3 * elapsed / args.steps, dom["launches"] // prof_steps))
    if dom.get("units"):
        perms_per_s = dom["units"] / max(dom["ms"], 1e-12) * 1e3
        rl["valu"] = {"what": "Poseidon2 permutations of this kernel class priced at %.0f VALU instructions each against the integer "
                              "issue peak: the bound that actually binds (the class moves a few per cent of the HBM peak)" % POSEIDON2_INSTR,
                      "permutations_per_s": perms_per_s, "achieved_Tops": perms_per_s * POSEIDON2_INSTR / 1e12, "peak_Tops": VALU_PEAK_TOPS,
                      "frac": perms_per_s * POSEIDON2_INSTR / 1e12 / VALU_PEAK_TOPS}
    result["roofline"] = rl
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_babybear(system.blob, trace, packed, proof.to_bytes(), rows)
    return result


def cpu_baseline_babybear(blob, trace, packed, gpu_proof, rows):
    """the oracle built for the BabyBear configuration (oracle/libms_oracle_bb.so), same input, same bytes"""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    load_oracle()
    import oracle_bb as ob

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(32, avail)
    ob.set_threads(cores)
    osys = ob.System(blob)
    best = None
    t_all = time.time()
    runs = 0
    while runs < 3 and (runs == 0 or time.time() - t_all < 20):
        t0 = time.perf_counter()
        want = osys.prove([trace], packed)
        dt = time.perf_counter() - t0
        runs += 1
        best = dt if best is None else min(best, dt)
        log("cpu baseline (BabyBear): proof %d took %.3f s" % (runs, dt))
    assert want == gpu_proof, "GPU proof differs from the oracle's"
    return {"value": rows / best, "unit": "rows/s", "cores": cores, "kind": "port",
            "sample": "oracle C++ restatement built with -DMSO_BABYBEAR (OpenMP, %d threads), same circuit, parameters and trace as the GPU "
                      "step (2^%d rows), best of %d proofs, %.3f s/proof, bytes identical to the GPU proof" % (
                          cores, rows.bit_length() - 1, runs, best)}


# ------------------------------------------------------------------------------------------------ N > 1
def timed_steps(args, ctx, torch, dist, step, sync_extra, dev):
    """W warm-up steps (the first one profiled), then exactly K steps between barrier + synchronize; max over ranks"""
    rank = dist.get_rank()

    def sync_all():
        ctx.sync()
        sync_extra()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    proof, dominant = profile_first_step(ctx, step, rank)
    for _ in range(max(args.warmup, 1) - 1):
        proof = step()
    ctx.set_profile([dominant])
    ctx.reset_stats()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dom = ctx.kernel_stats()[dominant]
    ctx.set_profile([])
    return proof, float(tmax.item()), dominant, dom


def replicas_leg(args, pkg, fe, ctx, torch, dist, mgpu, rank, local_rank, traces, claims):
    """one independent [ByteTable, U32Add] proof per rank per step (host-resident witnesses, as at N = 1); the only exchange
    is an all_gather of each rank's three commitments, folded into a joint digest on rank 0"""
    world = dist.get_world_size()
    system = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    packed = fe.pack_claims(claims)
    witness = system.host_witness(traces, packed)
    dev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")
    digests = []
    state = {"g": None}

    def step():
        # the commitments of proof k are gathered by a worker thread while proof k + 1 runs (CommitmentGatherer);
        # sync waits for all of them, so the timed region contains every collective it started
        proof = system.prove_multiple_claims(witness)
        blob = mgpu.commitments_of(proof.to_bytes(), 2)
        if state["g"] is None:
            state["g"] = mgpu.CommitmentGatherer(len(blob), dev if args.backend == "nccl" else None,
                                                 on_gathered=lambda allc: digests.append(mgpu.joint_digest(allc)))
        state["g"].submit(blob)
        return proof

    def sync_extra():
        if state["g"] is not None:
            state["g"].finish()

    proof, elapsed, dominant, dom = timed_steps(args, ctx, torch, dist, step, sync_extra, dev)
    if state["g"] is not None:
        state["g"].close()
    rows = witness.rows
    if rank == 0:
        log("replicas: %.3f ms per step (%d proofs per step); last joint digest %s" % (
            1e3 * elapsed / args.steps, world, digests[-1].hex() if digests else "-"))
    return {
        "what": "one independent [ByteTable, U32Add @ 2^%d] proof per GPU per step, host-resident witnesses, the three commitments of "
                "every proof all-gathered (RCCL) and digested on rank 0" % args.log_adds,
        "value": rows * world * args.steps / elapsed, "unit": "rows/s", "ms_per_step": 1e3 * elapsed / args.steps,
        "rows_per_proof": rows, "proofs_per_step": world, "proof_bytes": len(proof.to_bytes()),
    }, dominant, dom, rows


def joint_preflight(args, pkg, fe, ctx, torch, dist, sharded, system, comm, owners, rank, world, dev, wd):
    """Before anything is timed: the joint proof of the SAME system at 2^12 additions per rank must have exactly the bytes of
    the single-GPU proof of the full system (System::prove_multiple_claims on rank 0, which holds every trace at this size)
    and be accepted by the verifier. First contact with several GPUs then fails HERE, with a reason, instead of as a
    rejected proof after the measurement or as a hang in the middle of it."""
    if wd is not None:
        wd.leg = "joint proof, pre-flight at 2^12 additions per rank"
    k = min(args.log_adds, 12)
    traces, claims = fe.multi_u32_add_witness(world, 1 << k)  # deterministic: every rank builds the same small witness
    packed = fe.pack_claims(claims)
    mine = [t if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
    remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
    got = system.prove_sharded(system.host_witness(mine, packed, remote_heights=remote), comm, owners).to_bytes()
    again = system.prove_sharded(system.witness(mine, packed, remote_heights=remote), comm, owners).to_bytes()
    ok, why = 1, ""
    if again != got:
        ok, why = 0, "rank %d: the joint proof from a device-resident witness differs from the host-resident one" % rank
    if rank == 0 and ok:
        want = system.prove_multiple_claims(system.witness(traces, packed)).to_bytes()
        if got != want:
            first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
            ok, why = 0, "joint proof (%d bytes) differs from the single-GPU proof (%d bytes) of the same system from byte %d" % (len(got), len(want), first)
        elif system.verify_multiple_claims(packed, got) != 0:
            ok, why = 0, "the verifier rejects the pre-flight proof"
    # every rank must hold the same bytes; every rank learns the verdict
    digest = np.frombuffer(hashlib.sha256(got).digest(), dtype=np.uint8).astype(np.int64)
    flags = torch.tensor(np.concatenate([[ok], digest]), dtype=torch.int64, device=dev)
    parts = [torch.empty_like(flags) for _ in range(world)]
    dist.all_gather(parts, flags)
    parts = [p.cpu().numpy() for p in parts]
    all_ok = all(int(p[0]) == 1 for p in parts)
    same = all((p[1:] == parts[0][1:]).all() for p in parts)
    info = {"log_adds": k, "ok": bool(all_ok and same), "proof_bytes": len(got), "proof_sha256": hashlib.sha256(got).hexdigest(),
            "what": "joint proof of [ByteTable, U32Add x %d] at 2^%d additions per rank == System::prove_multiple_claims of the full system "
                    "on rank 0, byte for byte; same bytes on every rank; verifier accepts" % (world, k)}
    if not all_ok or not same:
        info["error"] = why or ("the ranks hold different proof bytes" if not same else "another rank reported a mismatch")
    if rank == 0:
        log("pre-flight: joint proof at 2^%d additions per rank %s the single-GPU proof of the same system (%d bytes, sha256 %s)" % (
            k, "==" if info["ok"] else "DIFFERS FROM", len(got), info["proof_sha256"][:16]))
    if not info["ok"]:
        raise PreflightFailed(info)
    return info


class PreflightFailed(RuntimeError):
    def __init__(self, info):
        super().__init__(info.get("error", "pre-flight failed"))
        self.info = info


def joint_leg(args, pkg, fe, ctx, torch, dist, rank, local_rank, traces, claims, wd=None):
    """ONE proof of [ByteTable, U32Add x N] by all ranks per step (ms_prove_sharded); setup (untimed, like criterion's setup
    closure): the byte table's multiplicities are the sum over ranks (plain integer counts) and every rank holds all
    claims, which the transcript absorbs in order"""
    import importlib

    sharded = importlib.import_module("multi_stark_amd.sharded")
    world = dist.get_world_size()
    dev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")
    system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(world))
    byte = torch.from_numpy(traces[0].astype(np.int64)).to(dev)
    dist.all_reduce(byte, op=dist.ReduceOp.SUM)
    mine_claims = torch.from_numpy(np.ascontiguousarray(claims).view(np.int64)).to(dev)
    parts = [torch.empty_like(mine_claims) for _ in range(world)]
    dist.all_gather(parts, mine_claims)
    packed = fe.pack_claims(np.concatenate([p.cpu().numpy().view(np.uint64) for p in parts], axis=0))
    del parts
    owners = sharded.u32_add_owners(world)
    tr = [byte.cpu().numpy().astype(np.uint64)] + [traces[1] if k == rank else None for k in range(world)]
    remote = {1 + k: traces[1].shape[0] for k in range(world) if k != rank}
    # host-resident, as at N = 1: every step uploads this rank's trace (and the byte table) and its slice of the claims
    witness = system.host_witness(tr, packed, remote_heights=remote)
    transport = args.transport if args.backend == "nccl" else "torch"
    if transport == "rccl":
        # bootstrap only: rank 0 draws the RCCL id, torch.distributed hands it round; the exchanges themselves are C
        box = [sharded.RcclComm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = sharded.RcclComm(ctx, box[0], rank, world)
    else:
        comm = sharded.TorchComm(local_rank)
    rows = 256 + world * traces[1].shape[0]
    preflight = joint_preflight(args, pkg, fe, ctx, torch, dist, sharded, system, comm, owners, rank, world, dev, wd)

    def step():
        return system.prove_sharded(witness, comm, owners)

    if wd is not None:
        wd.leg = "joint proof, first full-size proof (fills the pool, opens the transport's channels)"
    proof = step()  # fills the pool, opens the RCCL channels
    if wd is not None:
        wd.leg = "joint proof, warm-up and timed steps"
    comm.bytes_moved = 0
    proof, elapsed, dominant, dom = timed_steps(args, ctx, torch, dist, step, lambda: None, dev)
    moved = comm.bytes_moved // (args.steps + max(args.warmup, 1))
    stage = system.prove_sharded(witness, comm, owners, want_times=True).stage_ms
    sha = hashlib.sha256(proof.to_bytes()).hexdigest()
    # untimed: the joint proof must be accepted by the library's verifier (System::verify_multiple_claims on the bytes); a
    # wrong exchange on first contact with several GPUs must not pass as a measurement
    verdict = system.verify_multiple_claims(packed, proof.to_bytes()) if rank == 0 else 0
    if rank == 0:
        log("joint proof: verifier verdict %d (0 = accepted)" % verdict)
    if rank == 0:
        log("joint proof: %.3f ms per proof, %d rows, transport %s, sha256 %s" % (1e3 * elapsed / args.steps, rows, transport, sha[:16]))
    info = {
        "what": "ONE proof of [ByteTable, U32Add x %d] by %d ranks (ms_prove_sharded): row-range all-to-all per commitment, roots / "
                "totals / opened values / reduced openings all-gathered" % (world, world),
        "value": rows * args.steps / elapsed, "unit": "rows/s", "ms_per_step": 1e3 * elapsed / args.steps,
        "rows_per_proof": rows, "proof_bytes": len(proof.to_bytes()), "proof_sha256": sha, "transport": transport,
        "bytes_exchanged_per_rank_per_proof": moved, "stage_ms": {k: round(v, 3) for k, v in stage.items()},
        "verified": verdict == 0, "preflight": preflight,
    }
    if verdict != 0:
        info["error"] = "the joint proof is REJECTED by the verifier (code %d): the figures of this leg are void" % verdict
    return info, dominant, dom, rows


def multi_gpu(args, pkg, fe, ctx, torch, dist, rank, local_rank, n_gpus):
    import importlib

    mgpu = importlib.import_module("multi_stark_amd.distributed")
    a0, b0 = mgpu.rank_seeds(rank)  # per-rank seeds; rank 0 is exactly the reference's bench witness
    t = time.time()
    traces, claims = fe.u32_add_bench_witness(1 << args.log_adds, a0, b0)
    log("[rank %d] witness generated in %.1fs" % (rank, time.time() - t))
    done = {}

    def partial():
        # what rank 0 can still report when a later leg hangs
        if "replicas" in done:
            info = done["replicas"][0]
            line = base_line(args, n_gpus, info["value"], info["ms_per_step"])
            line["config"] = {"workload": WORKLOAD % (args.log_adds, "proof", HOST_RESIDENT), "parallelism": "1 proof per GPU (replicas)"}
            line["replicas"] = info
            return line
        return base_line(args, n_gpus, None, None)

    wd = Watchdog(rank, "the multi-GPU bench", args.primary_timeout, partial, ctx)
    joint_primary = not args.replicas_primary

    def run_joint():
        try:
            done["joint"] = joint_leg(args, pkg, fe, ctx, torch, dist, rank, local_rank, traces, claims, wd)
            return None
        except PreflightFailed as e:
            # nothing of the joint leg is timed on a prover whose bytes are wrong: report what there is, with the reason
            wd.finish()
            line = partial()
            line["error"] = "joint proof pre-flight failed: %s" % e
            line["preflight"] = e.info
            return line
        except Exception as e:  # noqa: BLE001
            # a failing exchange (the library aborts its communicator, so the peers fail too instead of hanging): say what and
            # where, keep the figures already measured, end with a non-zero status - and without the closing barrier, which
            # a rank that failed earlier would never join
            where = wd.where()
            wd.finish()
            log("[rank %d] joint proof failed in %s: %r" % (rank, where, e))
            line = partial()
            line["error"] = "joint proof failed (%s): %s" % (where, e)
            if rank == 0:
                print(json.dumps(line), flush=True)
            os._exit(4)

    def run_replicas():
        wd.leg = "replicas (one independent proof per rank)"
        done["replicas"] = replicas_leg(args, pkg, fe, ctx, torch, dist, mgpu, rank, local_rank, traces, claims)

    # The PRIMARY leg runs first: a secondary leg that hangs or fails on first contact with several GPUs must not take the
    # figure the run is for with it.
    if joint_primary:
        failed = run_joint()
        if failed is not None:
            return failed
    else:
        run_replicas()
    wd.finish()
    prim = done["joint"] if joint_primary else done["replicas"]
    info, dominant, dom, rows = prim
    result = base_line(args, n_gpus, info["value"], info["ms_per_step"])
    result["config"] = {
        "workload": WORKLOAD % (args.log_adds, "rank" if joint_primary else "proof",
                                "each rank's witness (its trace and all claims) in pinned host memory at step start: the rank uploads its trace and "
                                "its slice of the claims inside the step, as at N = 1" if joint_primary else HOST_RESIDENT),
        "rows_per_proof": rows,
        "proof_bytes": info["proof_bytes"],
        "parallelism": ("one joint proof over %d GPUs (ms_prove_sharded, transport %s)" % (n_gpus, info.get("transport"))) if joint_primary
                       else "1 proof per GPU (replicas)",
        "whole_path_alg_GBps": ALG_BYTES_PER_ROW * rows * (1 if joint_primary else n_gpus) / (info["ms_per_step"] / 1e3) / 1e9,
    }
    devs = [None] * n_gpus
    dist.all_gather_object(devs, int(local_rank))
    result["config"]["ranks"] = {"n": n_gpus, "devices": devs, "transport": info.get("transport") if joint_primary else None,
                                 "rccl_world": dist.get_world_size() if args.backend == "nccl" else None,
                                 "launcher": "torch.distributed.run (one process per GPU), backend %s" % args.backend}
    if joint_primary:
        result["config"]["stage_ms"] = info["stage_ms"]
        result["config"]["bytes_exchanged_per_rank_per_proof"] = info["bytes_exchanged_per_rank_per_proof"]
        result["config"]["proof_sha256"] = info["proof_sha256"]
        result["config"]["verified"] = info["verified"]
        if "error" in info:
            result["error"] = info["error"]
    result["roofline"] = roofline_of(dominant, dom, full_size=False)
    if joint_primary:
        result["config"]["preflight"] = info.get("preflight")
    # ---- the secondary leg, behind a watchdog of its own that prints the primary line (status 0) if the leg cannot finish
    want_secondary = (joint_primary and not args.no_replicas_leg) or (not joint_primary and not args.no_joint_leg)
    if want_secondary and not result.get("error"):
        wd2 = Watchdog(rank, "the secondary leg of the multi-GPU bench", min(args.primary_timeout, 240.0), lambda: result, ctx, soft=True)
        try:
            if joint_primary:
                wd2.leg = "replicas (one independent proof per rank)"
                done["replicas"] = replicas_leg(args, pkg, fe, ctx, torch, dist, mgpu, rank, local_rank, traces, claims)
                result["replicas"] = done["replicas"][0]
            else:
                wd2.leg = "joint proof (secondary)"
                done["joint"] = joint_leg(args, pkg, fe, ctx, torch, dist, rank, local_rank, traces, claims, wd2)
                result["joint_proof"] = done["joint"][0]
        except BaseException as e:  # noqa: BLE001  (the primary figures stand; the failure is part of the record)
            log("[rank %d] secondary leg failed: %r" % (rank, e))
            result["secondary_leg_error"] = "secondary leg failed: %r" % (e,)
            wd2.finish()
            if rank == 0:
                print(json.dumps(result), flush=True)
            os._exit(0)  # (no closing barrier: a rank that failed inside a collective would never join it)
        wd2.finish()
    if not args.no_cpu_baseline and rank == 0:
        # the same CPU leg as at N = 1 (one [ByteTable, U32Add] proof at 2^cpu-log-adds additions on this box's host cores): rows/s of
        # the restatement does not depend on how many adders the system holds. The other ranks wait at the closing barrier.
        one = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
        result["cpu_baseline"] = cpu_baseline(fe, one.blob, args.cpu_log_adds)
        result["cpu_baseline"]["sample"] += "; measured on rank 0's host cores while the other ranks idle"
        del one
    return result


def fail_line(args, why, code=2):
    """a run that cannot measure what the command names: say why (stderr and the JSON line) and end with a non-zero status"""
    log("bench.py: " + why)
    line = base_line(args, args.gpus, None, None)
    line["error"] = why
    print(json.dumps(line), flush=True)
    sys.exit(code)


# ------------------------------------------------------------------------------------------------ N > 1, one process
def local_multi_gpu(args):
    """`python3 bench.py --gpus N` without a launcher: ONE joint proof of [ByteTable, U32Add x N] per step by N thread ranks of
    this process, rank k on device k with its own ms_ctx, exchanges on the library's in-process transport (ms_comm_local_*:
    the receiving rank pulls its blocks out of the peers' buffers - peer copies over xGMI - ordered by HIP events). This is the
    reference's own process model (one process with a thread pool driving the box, Cargo.toml:45) and needs no rendezvous, no
    RCCL and no torch. Fewer than N visible devices is an ERROR (exit status 2), never a smaller measurement; --share-devices
    turns that into a labelled rehearsal. Timing: the ranks meet at a barrier, every rank synchronises its device, K steps,
    synchronise, barrier; the step time is the MAX over ranks."""
    import importlib

    pkg = load_package()
    fe = pkg.frontend
    sharded = importlib.import_module("multi_stark_amd.sharded")
    mgpu_seeds = lambda r: (0xDEADBEEF ^ ((r * 0x9E3779B9) & 0xFFFFFFFF), 0xCAFEBABE ^ ((r * 0x85EBCA6B) & 0xFFFFFFFF))  # noqa: E731
    N = args.gpus
    if N & (N - 1):
        fail_line(args, "the joint proof needs a power-of-two number of ranks (--gpus %d)" % N)
    ndev = pkg.device_count()
    if ndev < N and not args.share_devices:
        fail_line(args, "--gpus %d needs %d HIP devices, this process sees %d: refusing to measure fewer GPUs than the command names "
                        "(--share-devices rehearses the N-rank code path on the devices there are)" % (N, N, ndev))
    if ndev < 1:
        fail_line(args, "no HIP device visible")
    devices = [k % ndev for k in range(N)] if ndev < N else list(range(N))
    shared = len(set(devices)) < N
    log("in-process ranks: %d thread ranks on devices %s%s" % (N, devices, " (REHEARSAL: ranks share devices)" if shared else ""))
    num_adds = 1 << args.log_adds
    t = time.time()
    # setup (untimed, criterion's setup closure): rank k's witness from its own seeds; the byte table's multiplicities are the
    # sum over ranks; every rank holds all claims (the transcript absorbs them in order)
    byte = np.zeros((256, 1), dtype=np.uint64)
    adders, claims = [], []
    for r in range(N):
        (bt, add), cl = fe.u32_add_bench_witness(num_adds, *mgpu_seeds(r))
        byte += bt
        adders.append(add)
        claims.append(cl)
    packed = fe.pack_claims(np.concatenate(claims, axis=0))
    owners = sharded.u32_add_owners(N)
    rows = 256 + N * adders[0].shape[0]
    log("witness generated in %.1fs: %d adders x 2^%d rows, %d claims" % (time.time() - t, N, args.log_adds, len(packed[0]) - 1))
    # pre-flight reference (before any rank starts): the single-GPU proof of the same system at 2^12 additions per rank
    k = min(args.log_adds, 12)
    ptr, pcl = fe.multi_u32_add_witness(N, 1 << k)
    ppacked = fe.pack_claims(pcl)
    ctx0 = pkg.Context(devices[0])
    sys0 = pkg.System.new(ctx0, fe.bench_params(), fe.multi_u32_add_system_inputs(N))
    want_small = sys0.prove_multiple_claims(sys0.witness(ptr, ppacked)).to_bytes()
    small_ok = sys0.verify_multiple_claims(ppacked, want_small) == 0
    del sys0
    ctx0.trim()
    del ctx0
    barrier = threading.Barrier(N)
    out = {}
    ctxs = [None] * N
    state = {"leg": "setup"}

    def partial():
        line = base_line(args, N, None, None)
        if "replicas" in out:
            line["replicas"] = out["replicas"]
        return line

    class _Progress:  # what the watchdog prints: the exchange rank 0 is in
        def comm_progress(self):
            return ctxs[0].comm_progress() if ctxs[0] is not None else ("", 0, False)

    wd = Watchdog(0, "the multi-GPU bench (thread ranks)", args.primary_timeout, partial, _Progress())

    def timed(ctx, step, rank):
        """W warm-up steps (the first one profiled), then exactly K steps between barrier + device synchronisation"""
        proof, dominant = profile_first_step(ctx, step, rank)
        for _ in range(max(args.warmup, 1) - 1):
            proof = step()
        ctx.set_profile([dominant])
        ctx.reset_stats()
        ctx.sync()
        barrier.wait()
        t0 = time.perf_counter()
        per = []
        for _ in range(args.steps):
            t1 = time.perf_counter()
            proof = step()
            per.append(1e3 * (time.perf_counter() - t1))
        ctx.sync()
        barrier.wait()  # every rank has synchronised its device: the slowest rank closes the region
        elapsed = time.perf_counter() - t0
        dom = ctx.kernel_stats()[dominant]
        ctx.set_profile([])
        return proof, elapsed, dominant, dom, per

    def rank_body(rank, group):
        ctx = ctxs[rank] = pkg.Context(devices[rank])
        res = {}
        # ---- the joint proof
        system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(N))
        comm = group.comm(ctx, rank)
        try:
            if rank == 0:
                wd.leg = "joint proof, pre-flight at 2^%d additions per rank" % k
            mine = [t_ if owners[i] in (-1, rank) else None for i, t_ in enumerate(ptr)]
            remote = {i: ptr[i].shape[0] for i in range(len(ptr)) if owners[i] not in (-1, rank)}
            got = system.prove_sharded(system.host_witness(mine, ppacked, remote_heights=remote), comm, owners).to_bytes()
            again = system.prove_sharded(system.witness(mine, ppacked, remote_heights=remote), comm, owners).to_bytes()
            res["preflight"] = (got == want_small, again == got, hashlib.sha256(got).hexdigest(), len(got))
            barrier.wait()
            if not (got == want_small and again == got):
                return res  # (every rank leaves here or none does: all hold the same bytes or the check below reports it)
            tr = [byte] + [adders[r] if r == rank else None for r in range(N)]
            remote = {1 + r: adders[r].shape[0] for r in range(N) if r != rank}
            witness = system.host_witness(tr, packed, remote_heights=remote)
            if rank == 0:
                wd.leg = "joint proof, first full-size proof (fills the pool, enables peer access)"
            step = lambda: system.prove_sharded(witness, comm, owners)  # noqa: E731
            step()
            moved0 = comm.bytes_moved
            if rank == 0:
                wd.leg = "joint proof, warm-up and timed steps"
            proof, elapsed, dominant, dom, per = timed(ctx, step, rank)
            moved = (comm.bytes_moved - moved0) // (args.steps + max(args.warmup, 1))
            stage = system.prove_sharded(witness, comm, owners, want_times=True).stage_ms
            data = proof.to_bytes()
            res["joint"] = {"elapsed": elapsed, "per_step_ms": per, "dominant": dominant, "dom": dom, "moved": moved, "stage": stage,
                            "sha": hashlib.sha256(data).hexdigest(), "bytes": len(data),
                            "verdict": system.verify_multiple_claims(packed, data) if rank == 0 else 0}
            del witness
        finally:
            comm.close()
        # ---- secondary leg, behind the primary one: one independent [ByteTable, U32Add] proof per rank per step
        if not args.no_replicas_leg and "joint" in res:
            if rank == 0:
                wd.leg = "replicas (one independent proof per rank)"
            one = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
            (bt, add), cl = fe.u32_add_bench_witness(num_adds, *mgpu_seeds(rank))
            opacked = fe.pack_claims(cl)
            ow = one.host_witness([bt, add], opacked)
            proof, elapsed, _, _, _ = timed(ctx, lambda: one.prove_multiple_claims(ow), rank)
            res["replicas"] = (elapsed, len(proof.to_bytes()), ow.rows, hashlib.sha256(proof.to_bytes()).hexdigest())
            del ow, one
            ctx.trim()
        return res

    group = sharded.LocalGroup(N)
    try:
        results = group.run(rank_body)
    except Exception as e:  # noqa: BLE001  (a failing rank aborts the group: its peers return with an error, nobody hangs)
        where = wd.where()
        wd.finish()
        line = partial()
        line["error"] = "joint proof by thread ranks failed (%s): %s" % (where, e)
        log(line["error"])
        return line
    finally:
        group.close()
    wd.finish()
    ranks_info = {"n": N, "devices": devices, "transport": "local (ms_comm_local_*: N thread ranks of one process, peer copies ordered by HIP events)",
                  "rccl_world": None, "launcher": "none (in-process)", "devices_visible": ndev, "ranks_share_devices": shared}
    pre = [r["preflight"] for r in results]
    preflight = {"log_adds": k, "proof_bytes": pre[0][3], "proof_sha256": pre[0][2],
                 "ok": bool(small_ok and all(p[0] and p[1] for p in pre) and len({p[2] for p in pre}) == 1),
                 "what": "joint proof of [ByteTable, U32Add x %d] at 2^%d additions per rank == System::prove_multiple_claims of the full system on "
                         "device %d, byte for byte, from host- and device-resident witnesses; same bytes on every rank; verifier accepts" % (N, k, devices[0])}
    replicas = None
    if not args.no_replicas_leg and all("replicas" in r for r in results):
        el = max(r["replicas"][0] for r in results)
        replicas = {"what": "one independent [ByteTable, U32Add @ 2^%d] proof per rank per step (thread ranks, host-resident witnesses)" % args.log_adds,
                    "value": results[0]["replicas"][2] * N * args.steps / el, "unit": "rows/s", "ms_per_step": 1e3 * el / args.steps,
                    "rows_per_proof": results[0]["replicas"][2], "proofs_per_step": N, "proof_bytes": results[0]["replicas"][1]}
        out["replicas"] = replicas
    if not preflight["ok"]:
        line = partial()
        line["error"] = "joint proof pre-flight failed: the joint proof differs from the single-GPU proof of the same system (or between ranks)"
        line["preflight"] = preflight
        line["config"] = {"ranks": ranks_info}
        return line
    joint = [r["joint"] for r in results]
    elapsed = max(j["elapsed"] for j in joint)
    same = len({j["sha"] for j in joint}) == 1
    j0 = joint[0]
    per = np.max(np.array([j["per_step_ms"] for j in joint]), axis=0)  # a step ends when its slowest rank has the bytes
    result = base_line(args, N, rows * args.steps / elapsed, 1e3 * elapsed / args.steps)
    result["config"] = {
        "workload": WORKLOAD % (args.log_adds, "rank", "each rank's witness (its trace and all claims) in pinned host memory at step start: the rank uploads "
                                "its trace and its slice of the claims inside the step, as at N = 1"),
        "rows_per_proof": rows, "proof_bytes": j0["bytes"],
        "parallelism": "one joint proof over %d %s (ms_prove_sharded, transport local)" % (N, "thread ranks SHARING %d device(s): rehearsal" % len(set(devices)) if shared else "GPUs"),
        "ranks": ranks_info,
        "whole_path_alg_GBps": ALG_BYTES_PER_ROW * rows / (elapsed / args.steps) / 1e9,
        "stage_ms": {k_: round(v, 3) for k_, v in j0["stage"].items()},
        "step_ms_min_median_max": [round(float(np.min(per)), 3), round(float(np.median(per)), 3), round(float(np.max(per)), 3)],
        "bytes_exchanged_per_rank_per_proof": j0["moved"], "proof_sha256": j0["sha"], "verified": j0["verdict"] == 0 and same,
        "preflight": preflight,
    }
    if j0["verdict"] != 0 or not same:
        result["error"] = ("the joint proof is REJECTED by the verifier (code %d)" % j0["verdict"]) if j0["verdict"] else "the ranks hold different proof bytes"
    result["roofline"] = roofline_of(j0["dominant"], j0["dom"], full_size=False)
    if not args.no_cpu_baseline:
        ctx = pkg.Context(devices[0])
        one = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
        result["cpu_baseline"] = cpu_baseline(fe, one.blob, args.cpu_log_adds)
        result["cpu_baseline"]["sample"] += "; measured after the GPU legs, all ranks idle"
        del one, ctx
    if replicas is not None:
        result["replicas"] = replicas
    return result


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if world > 1 and world != args.gpus:
        # never measure another number of GPUs than the command names
        fail_line(args, "launched with WORLD_SIZE=%d but --gpus %d: the two must agree" % (world, args.gpus))
    if world > 1 and args.transport == "local":
        fail_line(args, "--transport local runs the N ranks as threads of ONE process: start it without a launcher (python3 bench.py --gpus N)")
    if world == 1 and args.gpus > 1:
        if args.transport in ("rccl", "torch"):
            fail_line(args, "--transport %s needs one process per GPU: launch with python -m torch.distributed.run --nproc-per-node %d "
                            "(or leave --transport out: the ranks then run as threads of this process)" % (args.transport, args.gpus))
        if args.config == "babybear":
            fail_line(args, "--config babybear is a single-GPU measurement")
        # no launcher environment: ONE process drives the N devices itself, as the reference's prover does
        result = local_multi_gpu(args)
        print(json.dumps(result), flush=True)
        if result.get("error"):
            sys.exit(4)
        return
    if args.transport is None:
        args.transport = "rccl"
    n_gpus = world if world > 1 else 1
    # rehearsal of the collective path with a single rank (RCCL initialises and gathers with world size 1)
    force_dist = world == 1 and bool(os.environ.get("MSAMD_BENCH_FORCE_DIST")) and "RANK" in os.environ

    torch = None
    try:
        import torch as _torch

        torch = _torch
    except Exception as e:  # torch is plumbing only (barrier / synchronize / bootstrap); never needed for the proof itself
        if n_gpus > 1 or force_dist:
            raise
        log("torch unavailable (%s): using the library's own stream synchronisation" % e)
    dist = None
    if n_gpus > 1 or force_dist:
        import torch.distributed as _dist

        dist = _dist
        ndev = torch.cuda.device_count()
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(ndev, 1)  # rehearsal: ranks may share a device
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=args.backend)

    pkg = load_package()
    fe = pkg.frontend
    ctx = pkg.Context(local_rank)
    if args.config == "babybear":
        if dist is not None:
            raise SystemExit("--config babybear is a single-GPU measurement")
        result = babybear(args, pkg, fe, ctx, torch)
    elif dist is None:
        result = single_gpu(args, pkg, fe, ctx, torch)
        if not args.no_config4 and not args.hbm_resident and args.log_adds == 20:
            # BASELINE config 4 (the reference's second StarkGenericConfig) as a secondary object of the same line, so that the
            # run that records config 2 also records it; its full line (roofline, CPU baseline) is `--config babybear`
            try:
                import copy

                a4 = copy.copy(args)
                a4.steps, a4.warmup, a4.no_cpu_baseline = max(3, min(args.steps, 10)), 2, True
                r4 = babybear(a4, pkg, fe, ctx, torch)
                result["config4_babybear"] = {
                    "what": "BASELINE config 4, secondary leg of this run: " + r4["config"]["workload"],
                    "value": r4["value"], "unit": r4["unit"], "ms_per_step": r4["ms_per_step"], "steps": a4.steps,
                    "hbm_resident_ms": r4["config"]["hbm_resident_ms"], "stage_ms": r4["config"]["stage_ms"],
                    "proof_bytes": r4["config"]["proof_bytes"], "verified": r4["config"]["verified"],
                    "step_ms_min_median_max": r4["config"]["step_ms_min_median_max"],
                    "note": "timed without HIP events; the dominant kernel class (%s) is timed in a separate pass (`--config babybear` "
                            "prints the full line with roofline and CPU baseline)" % r4["roofline"]["kernel"]}
                log("config 4 (BabyBear / Poseidon2) secondary leg: %.3f ms per step, %.3f ms HBM-resident" % (
                    r4["ms_per_step"], r4["config"]["hbm_resident_ms"] or float("nan")))
            except Exception as e:  # noqa: BLE001  (a secondary figure must not void the run, but its failure is part of the record)
                log("config 4 secondary leg failed: %r" % (e,))
                result["config4_babybear"] = {"error": "config 4 secondary leg failed: %r" % (e,)}
    else:
        result = multi_gpu(args, pkg, fe, ctx, torch, dist, rank, local_rank, n_gpus)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if result.get("error"):
        sys.exit(4)  # a rejected proof is a failed run, whatever was timed


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run
    separately on this same command, corrected as MI355X_MICROARCH.md prescribes; tools/traffic_from_pmc.py)."""
    for name in TRAFFIC_FILES:
        try:
            return json.load(open(os.path.join(ROOT, "profiles", name)))[kernel]["hbm_bytes_per_launch"], name
        except Exception:
            continue
    return None, None


def native_oracle_dir():
    """The oracle compiled on THIS box with -march=native (the reference's .cargo/config.toml sets -Ctarget-cpu=native); the copy
    that travelled with the repository is built for x86-64-v3. None when it cannot be built here (no compiler): the caller
    then times the portable build and says so."""
    import subprocess

    if os.environ.get("MSAMD_BENCH_PORTABLE_ORACLE"):
        return None
    out = os.path.join(ROOT, "oracle", "_native")
    try:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ARCH=native", "OUT=_native"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL, timeout=300)
        return out if os.path.exists(os.path.join(out, "libms_oracle.so")) else None
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(fe, blob, log_adds):
    """The oracle (multi-threaded C++ restatement, kind "port") timed on this box's host cores over the SAME workload
    as the GPU step (2^log_adds additions; one proof is a few seconds); the thread count is chosen by a quick sweep
    on a 2^16 sample first. It is a reported baseline, not the thing measured or shipped."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    native = None
    if "oracle" not in sys.modules:  # (the library is chosen when the wrapper is first imported)
        native = native_oracle_dir()
        if native:
            os.environ["MSO_ORACLE_DIR"] = native
    log("cpu baseline: oracle built with -march=%s" % ("native (on this box)" if native else "x86-64-v3 (the build that travelled with the repository)"))
    oracle = load_oracle()
    osys = oracle.System(blob)
    t_all = time.time()
    # thread sweep on a small sample: the restatement's parallel loops stop scaling before the box's full thread count
    st, sc = fe.u32_add_bench_witness(1 << min(16, log_adds))
    sp = fe.pack_claims(sc)
    sweep = {}
    # never the box's full thread count: a one-GPU box shows 256 hardware threads but grants a share of about 16 CPUs, and
    # 256 OpenMP threads spinning at barriers on that share do not finish in minutes once the GPU runtime's helper
    # threads are also running (seen on this pool; the sweep peaks at 16-32 threads anyway)
    for cores in sorted({c for c in (8, 16, 32, 64) if c <= avail}):
        oracle.set_threads(cores)
        osys.prove(st, sp)  # warm-up (twiddle tables, thread pool)
        _, tm = osys.prove(st, sp, want_times=True)
        sweep[cores] = tm["total"]
        if time.time() - t_all > 8:
            break
    best_cores = min(sweep, key=sweep.get)
    log("cpu baseline: thread sweep %s -> %d threads" % ({k: round(v, 3) for k, v in sweep.items()}, best_cores))
    oracle.set_threads(best_cores)
    traces, claims = fe.u32_add_bench_witness(1 << log_adds)
    packed = fe.pack_claims(claims)
    rows = sum(t.shape[0] for t in traces)
    best, runs = None, 0
    while runs < 3 and (runs == 0 or time.time() - t_all < 24):
        _, tm = osys.prove(traces, packed, want_times=True)
        runs += 1
        best = tm["total"] if best is None else min(best, tm["total"])
        log("cpu baseline: proof %d at 2^%d additions took %.3f s" % (runs, log_adds, tm["total"]))
    return {
        "value": rows / best,
        "unit": "rows/s",
        "cores": best_cores,
        "kind": "port",
        "sample": "oracle C++ restatement (OpenMP, -O3 -march=%s; %d threads, the best of a sweep over %s at 2^%d additions), same circuit and "
                  "params as the GPU step, 2^%d additions per proof, best of %d proofs, %.3f s/proof (witness prep excluded; "
                  "%d host threads available)" % ("native" if native else "x86-64-v3", best_cores, sorted(sweep), min(16, log_adds), log_adds, runs, best, avail),
    }


if __name__ == "__main__":
    main()
