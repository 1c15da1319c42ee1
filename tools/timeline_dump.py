"""Per-kernel timeline of the last proof in a rocprofv3 kernel-trace CSV: start offset, gap before, duration (us)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "transpose_in" in r["Kernel_Name"]]
rows = rows[idx[-2]:]
t0 = int(rows[0]["Start_Timestamp"])
prev = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("msamd::(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    print("%8.1f  gap %6.1f  dur %7.1f  %-36s grid %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n[:36], r.get("Grid_Size_X", "")))
    prev = max(prev, e)
