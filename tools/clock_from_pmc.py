"""Effective shader clock and vector-ALU issue rate per kernel class from ONE rocprofv3 pass
   rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -- python3 bench.py --hbm-resident ...
clock = GRBM_GUI_ACTIVE / 8 / duration (the counter is summed over the 8 XCDs: MI355X_MICROARCH.md, "DVFS give-back"; the
quotient reads high on dispatches shorter than about 0.3 ms - dispatches under 100 us are left out). The VALU rate is
SQ_INSTS_VALU x 64 lanes / duration of the SAME dispatches, set against the issue peak at the nominal 2.4 GHz and at the
clock the card actually held (256 CUs x 4 SIMDs x 16 lanes x clock).
usage: clock_from_pmc.py counter_collection.csv > out.txt"""
import csv
import sys
from collections import defaultdict

CLASS = {"ntt8s_k<false": "ntt8s_dif", "ntt8s_k<true": "ntt8s_dit", "ntt12_k<false": "ntt12_dif", "ntt12_k<true": "ntt12_dit",
         "leaf_hash_single_k": "leaf_hash", "deep_reduce_k": "deep_reduce", "quotient_jit": "quotient", "compress3_k": "compress_layer",
         "stage2_terms_trace_jit": "stage2_terms", "inv_denoms_k": "inv_denoms", "bary_partial_batch_k": "bary", "subtree_k<true, false, true>": "fri_round"}
disp = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    d = disp[r["Dispatch_Id"]]
    d["name"] = r["Kernel_Name"]
    d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
CLASS.update({"stage2_write_k": "stage2_write", "claims_acc_k": "claims_acc", "chunk_cv_k": "claims_chunks", "fri_tail_k": "fri_tail",
              "subtree_k<false, true, false>": "tree_top", "transpose_in_br_k": "transpose"})
acc = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
short = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for d in disp.values():
    if "GRBM_GUI_ACTIVE" not in d:
        continue
    for pat, name in CLASS.items():
        if pat in d["name"]:
            a = (acc if d["dur"] >= 100e-6 else short)[name]
            if d["dur"] < 30e-6:
                break
            a[0] += 1
            a[1] += d["dur"]
            a[2] += d["GRBM_GUI_ACTIVE"]
            a[3] += d.get("SQ_INSTS_VALU", 0.0)
            break
print("dispatches of at least 100 us, HBM-resident proofs of the bench workload, under the counter pass")
print("%-16s %5s %10s %10s %10s %16s %14s" % ("class", "disp", "avg us", "clock GHz", "VALU T/s", "of 2.4 GHz peak", "of clock peak"))
for name in sorted(acc, key=lambda n: -acc[n][1]):
    n, t, g, v = acc[name]
    clock = g / 8.0 / t
    rate = 64.0 * v / t
    print("%-16s %5d %10.1f %10.3f %10.1f %16.3f %14.3f" % (name, n, 1e6 * t / n, clock / 1e9, rate / 1e12, rate / (256 * 4 * 16 * 2.4e9),
                                                           rate / (256 * 4 * 16 * clock)))
print()
print("dispatches of 30 to 100 us (the clock quotient reads high there: the nominal peak only)")
print("%-16s %5s %10s %10s %16s" % ("class", "disp", "avg us", "VALU T/s", "of 2.4 GHz peak"))
for name in sorted(short, key=lambda n: -short[n][1]):
    n, t, g, v = short[name]
    if n == 0:
        continue
    rate = 64.0 * v / t
    print("%-16s %5d %10.1f %10.1f %16.3f" % (name, n, 1e6 * t / n, rate / 1e12, rate / (256 * 4 * 16 * 2.4e9)))
