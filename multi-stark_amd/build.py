"""Build the gfx950 prover library in-tree: csrc/*.hip -> multi-stark_amd/libmstark_hip.so (hipcc cross-compiles
without a GPU). Objects are cached under csrc/build/ and rebuilt when a source or header is newer."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libmstark_hip.so")
SOURCES = ["ctx.hip", "ntt.hip", "hash.hip", "lookup.hip", "quotient.hip", "quotient_jit.hip", "open.hip", "outer.hip", "prover.hip", "verifier.hip", "witness_gen.hip", "capi.hip",
           "comm_rccl.hip", "comm_local.hip", "bb_kernels.hip", "bb_prover.hip", "pack_host.cpp"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]  # .inc: sources included by a .hip
    headers.append(os.path.join(os.path.dirname(HERE), "include", "mstark.h"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "mstark_bb.h"))
    hdr_mtime = max(os.path.getmtime(h) for h in headers)
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_mtime):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        flags = [f for f in FLAGS if not f.startswith("--offload-arch")] if src.endswith(".cpp") else FLAGS  # plain host C++
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-6000:]))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if jobs or force or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-6000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
