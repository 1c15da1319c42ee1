"""Host-resident step time by the narrow upload's settings (each setting in a fresh process: the pack pool is created once per
process): MSAMD_PACK_THREADS x MSAMD_PACK_CHUNKS, with and without the pulling kernel (MSAMD_NO_PULL).
  python tools/pack_sweep.py            (driver)
  python tools/pack_sweep.py child N    (one setting: N proofs, prints mean / min / median ms)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from __graft_entry__ import load_package

    n = int(sys.argv[2])
    pkg = load_package()
    fe = pkg.frontend
    ctx = pkg.Context(0)
    system = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    traces, claims = fe.u32_add_bench_witness(1 << 20)
    packed = fe.pack_claims(claims)
    w = system.host_witness(traces, packed)
    for _ in range(6):
        system.prove_multiple_claims(w)
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        system.prove_multiple_claims(w)
        ts.append(1e3 * (time.perf_counter() - t))
    dw = system.witness(traces, packed)
    for _ in range(3):
        system.prove_multiple_claims(dw)
    hs = []
    for _ in range(10):
        t = time.perf_counter()
        system.prove_multiple_claims(dw)
        hs.append(1e3 * (time.perf_counter() - t))
    print("mean %.3f min %.3f median %.3f max %.3f | hbm-resident median %.3f" % (np.mean(ts), min(ts), np.median(ts), max(ts), np.median(hs)), flush=True)
    sys.exit(0)

settings = []
for threads in (16, 24, 32, 48, 64):
    for chunks in (8, 16, 32):
        settings.append({"MSAMD_PACK_THREADS": str(threads), "MSAMD_PACK_CHUNKS": str(chunks)})
settings.append({"MSAMD_PACK_THREADS": "16", "MSAMD_PACK_CHUNKS": "8", "MSAMD_NO_PULL": "1"})
settings.append({"MSAMD_PACK_THREADS": "32", "MSAMD_PACK_CHUNKS": "16", "MSAMD_NO_PULL": "1"})
for extra in settings:
    env = dict(os.environ)
    env.update(extra)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", "40"], capture_output=True, text=True, env=env, timeout=600)
    print("%-70s %s" % (extra, r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)
