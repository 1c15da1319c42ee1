"""Kernels and memory copies of the last HOST-resident proof of a rocprofv3 --kernel-trace --memory-copy-trace run
(tools/trace_host.py): offsets from the first host-to-device copy of that proof."""
import csv
import glob
import sys

d = sys.argv[1]
k = list(csv.DictReader(open(glob.glob(d + "/*_kernel_trace.csv")[0])))
mc = glob.glob(d + "/*_memory_copy_trace.csv")  # (absent from a kernel-only trace: the memory-copy domain stretches the copies' gaps)
m = list(csv.DictReader(open(mc[0]))) if mc else []
ev = []
for r in k:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel " + r["Kernel_Name"].replace("msamd::(anonymous namespace)::", "").split("(")[0][:44]))
for r in m:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy   " + r.get("Direction", "")))
ev.sort()
# the last proof starts at the first H2D copy after the last gather_queries_k of the previous one
gq = [i for i, e in enumerate(ev) if "gather_queries_k" in e[2]]
i0 = next(i for i in range(gq[-2], len(ev)) if "HOST_TO_DEVICE" in ev[i][2] or "pull_widen_k" in ev[i][2])
t0 = ev[i0][0]
last = ev[gq[-1]][1]
n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 46
for s, e, n in ev[i0:i0 + n_show]:
    print("%9.1f  end %9.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
print("...")
print("last kernel of the proof (gather_queries_k) ends at %.1f us" % ((last - t0) / 1e3))
