"""A / B of environment switches on the bench workload: every setting gets its own process (the library reads its switches
once), the settings are run in turn `--reps` times so that drift of the box hits all of them alike.
usage: python tools/ab_env.py [--mode hbm|host] [--reps 2] base MSAMD_X=1 MSAMD_Y=2,MSAMD_Z=1 ...   ("base" = no switch set)"""
import argparse
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="hbm")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("settings", nargs="+")
    a = ap.parse_args()
    res = {s: [] for s in a.settings}
    for rep in range(a.reps):
        for s in a.settings:
            env = dict(os.environ)
            if s != "base":
                for kv in s.split(","):
                    k, v = kv.split("=", 1)
                    env[k] = v
            r = subprocess.run([sys.executable, os.path.join(HERE, "series_modes.py"), a.mode], env=env, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                print(s, "FAILED", r.stderr[-800:], flush=True)
                continue
            m = re.search(r"mean ([0-9.]+) median ([0-9.]+)", r.stdout)
            res[s].append((float(m.group(1)), float(m.group(2))))
            print("rep %d %-50s %s" % (rep, s, r.stdout.strip()), flush=True)
    print()
    for s in a.settings:
        if res[s]:
            print("%-50s median of medians %.3f ms   (medians: %s)" % (s, sorted(x[1] for x in res[s])[len(res[s]) // 2], " ".join("%.3f" % x[1] for x in res[s])))


if __name__ == "__main__":
    main()
