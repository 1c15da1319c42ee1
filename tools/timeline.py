"""From a rocprofv3 kernel-trace CSV: busy vs idle time of the last proof (gaps > 1.5 us listed by size)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last proof = from the last transpose_in_k of the big trace backwards: take kernels after the 2nd-to-last claims_words_k
idx = [i for i, r in enumerate(rows) if "transpose_in" in r["Kernel_Name"]]
start = idx[-2] if len(idx) >= 2 else 0
# find beginning of last proof: last pair of transpose_in (byte table + main)
rows = rows[start:]
t0 = int(rows[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("kernels %d span %.3f ms busy %.3f ms idle %.3f ms" % (len(rows), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
gaps = []
prev_end = int(rows[0]["End_Timestamp"])
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - prev_end
    if g > 1500:
        gaps.append((g, a["Kernel_Name"][:60], b["Kernel_Name"][:60]))
    prev_end = max(prev_end, int(b["End_Timestamp"]))
gaps.sort(reverse=True)
print("gaps > 1.5us: %d totalling %.3f ms" % (len(gaps), sum(g for g, _, _ in gaps) / 1e6))
for g, a, b in gaps[:25]:
    print("%8.1f us  after %-60s before %s" % (g / 1e3, a.replace("msamd::(anonymous namespace)::", ""), b.replace("msamd::(anonymous namespace)::", "")))
