"""Build profiles/<name>.json (HBM bytes per launch per kernel) from rocprofv3 --pmc passes.
usage: traffic_from_pmc.py FETCH.csv WRITE.csv out.json
FETCH_SIZE/WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE reports half of the bytes of wide coalesced streaming reads
(MI355X_MICROARCH.md §HBM); calibrated here on ntt12_k (in-place pass: bytes read == bytes written, and
FETCH_SIZE == WRITE_SIZE / 2 is what the counters show), so reads are doubled."""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    return acc, cnt


# algorithmic (compulsory) bytes per launch of the transform classes at the bench size (config 2: 42 columns x 2^22 rows x 16 B
# per pass, 3 launches per pass). A counter reading BELOW this is not saved traffic - every byte of an in-place pass has to
# be read and written once - but a counter artefact: the x 2 FETCH_SIZE correction is calibrated on 16 B / lane streaming loads
# (MI355X_MICROARCH.md, HBM) and other access widths are uncalibrated, and Infinity-Cache hits are counted. Such an entry
# carries a note and `below_compulsory: true`.
ALG = {"ntt12_dif": 939524096.0, "ntt8s_dif": 939524096.0}

fa, fc = load(sys.argv[1], "FETCH_SIZE")
wa, wc = load(sys.argv[2], "WRITE_SIZE")
CLASS = {"ntt8s_k<false": "ntt8s_dif", "ntt8s_k<true": "ntt8s_dit", "ntt12_k<false": "ntt12_dif", "ntt12_k<true": "ntt12_dit",
         "leaf_hash_k": "leaf_hash", "deep_reduce_k": "deep_reduce", "quotient_k": "quotient", "stage2_terms_k": "stage2_terms"}
out = {}
for k in fa:
    for pat, name in CLASS.items():
        if pat in k:
            n = max(fc[k], 1)
            fetch_kb, write_kb = fa[k] / n, wa.get(k, 0.0) / max(wc.get(k, 1), 1)
            out[name] = {"launches_sampled": n, "fetch_size_kib_raw": fetch_kb, "write_size_kib": write_kb,
                         "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
                         "note": "reads = 2 x FETCH_SIZE (gfx950 correction), writes = WRITE_SIZE"}
            if name in ALG:
                out[name]["alg_bytes_per_launch"] = ALG[name]
                if out[name]["hbm_bytes_per_launch"] < 0.98 * ALG[name]:
                    out[name]["below_compulsory"] = True
                    out[name]["note"] += ("; BELOW the compulsory %.1f MB of this launch: a counter artefact (the x 2 correction is calibrated on "
                                          "16 B / lane streaming loads, this kernel reads 128-byte strided runs; Infinity-Cache hits are counted), not "
                                          "saved traffic" % (ALG[name] / 1e6))
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
