"""Summarise a rocprofv3 --kernel-trace --stats CSV: per-kernel calls / total / average, sorted by total."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
print("%-78s %8s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-78s %8.1f %10.3f %10.2f %6s" % (r["Name"][:78], int(r["Calls"]) / div, float(r["TotalDurationNs"]) / 1e6 / div,
                                           float(r["AverageNs"]) / 1e3, r["Percentage"]))
