"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        c = r["Counter_Name"]
        acc[k][c] += float(r["Counter_Value"])
        cnt[k][c] += 1
names = sorted({c for k in acc for c in acc[k]})
print("%-60s %7s " % ("kernel", "disp") + " ".join("%16s" % n[:16] for n in names))
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    d = max(cnt[k].values())
    print("%-60s %7d " % (k.replace("msamd::(anonymous namespace)::", "")[:60], d) +
          " ".join("%16.4g" % (acc[k][n] / max(cnt[k][n], 1)) for n in names))
