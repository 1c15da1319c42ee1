"""Build profiles/<name>.json (VALU instructions per launch per kernel class) from a rocprofv3 --pmc SQ_INSTS_VALU pass.
usage: valu_from_pmc.py counter_collection.csv out.json
SQ_INSTS_VALU counts wave-level instructions; x 64 lanes = lane-operations, which is what the 39.4 T/s integer issue peak
of MI355X (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz) is stated in."""
import csv
import json
import sys
from collections import defaultdict

acc, cnt = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "SQ_INSTS_VALU":
        acc[r["Kernel_Name"]] += float(r["Counter_Value"])
        cnt[r["Kernel_Name"]] += 1
CLASS = {"ntt8s_k<false": "ntt8s_dif", "ntt8s_k<true": "ntt8s_dit", "ntt12_k<false": "ntt12_dif", "ntt12_k<true": "ntt12_dit",
         "leaf_hash_single_k": "leaf_hash", "deep_reduce_k": "deep_reduce", "quotient_jit": "quotient", "compress3_k": "compress_layer"}
ALG = {"ntt12_dif": 939524096.0, "ntt8s_dif": 939524096.0}  # algorithmic bytes per launch at the bench size (config 2)
out = {}
for k in acc:
    for pat, name in CLASS.items():
        if pat in k and name not in out:
            out[name] = {"launches_sampled": cnt[k], "valu_wave_insts_per_launch": acc[k] / cnt[k],
                         "valu_lane_ops_per_launch": 64.0 * acc[k] / cnt[k]}
            if name in ALG:
                out[name]["alg_bytes_per_launch"] = ALG[name]
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
