"""CPU tests: the C-ABI library loads without a GPU and exports every symbol include/mstark.h declares; the product
path fails loudly (no fallback) when no HIP device exists."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mstark.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported(pkg):
    assert os.path.exists(pkg.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(pkg.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for sym in declared:
        assert hasattr(L, sym), "include/mstark.h declares %s but the library does not export it" % sym
    assert sorted(pkg.exported_symbols()) == declared


def test_babybear_header_symbols_are_exported(pkg):
    """include/mstark_bb.h: the reference's second configuration (BabyBear / Poseidon2)"""
    txt = open(os.path.join(ROOT, "include", "mstark_bb.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    declared = sorted(set(re.findall(r"\b(msbb_[a-z0-9_]+)\s*\(", txt)))
    L = ctypes.CDLL(pkg.LIB_PATH)
    assert len(declared) >= 15
    for sym in declared:
        assert hasattr(L, sym), "include/mstark_bb.h declares %s but the library does not export it" % sym
    assert sorted(pkg.babybear.exported_symbols()) == declared


def test_rust_binding_lists_every_symbol():
    """bindings/rust/mstark_sys.rs (uncompiled source for the reference side) declares exactly the headers' entry points"""
    rs = open(os.path.join(ROOT, "bindings", "rust", "mstark_sys.rs")).read()
    declared = set(re.findall(r"pub fn ((?:ms|msbb)_[a-z0-9_]+)\(", rs))
    want = set(_declared_symbols())
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mstark_bb.h")).read(), flags=re.S)
    want |= set(re.findall(r"\b(msbb_[a-z0-9_]+)\s*\(", txt))
    assert declared == want, (sorted(want - declared), sorted(declared - want))


def _header_comm_members():
    txt = open(os.path.join(ROOT, "include", "mstark.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    body = re.search(r"typedef struct ms_comm \{(.*?)\} ms_comm;", txt, flags=re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"\w[\w\s\*]*\(\*(\w+)\)\(", decl)  # function pointer member
        if m:
            names.append(m.group(1))
        else:  # plain members, possibly several per declaration ("int32_t rank, world")
            names += [n.strip(" *") for n in decl.split(" ", 1)[1].split(",")]
    return names


def test_ms_comm_members_agree_across_bindings(pkg):
    """the callback table of ms_prove_sharded has the same members, in the same order, in include/mstark.h, in the Rust binding
    and in the ctypes mirror - a binding that ends early makes the library read past the host's struct (ms_comm.size guards
    that at run time; this guards it at review time)"""
    import importlib

    header = _header_comm_members()
    assert header[:4] == ["size", "rank", "world", "user"] and header[-1] == "abort" and len(header) == 13, header
    rs = open(os.path.join(ROOT, "bindings", "rust", "mstark_sys.rs")).read()
    body = re.search(r"pub struct ms_comm \{(.*?)\n\}", rs, flags=re.S).group(1)
    body = re.sub(r"///[^\n]*", "", body)
    rust = re.findall(r"pub (\w+):", body)
    assert rust == header, (rust, header)
    sharded = importlib.import_module("multi_stark_amd.sharded")
    py = [f[0] for f in sharded.MsComm._fields_]
    assert py == header, (py, header)
    # and the ctypes layout is the C layout: one u32 + two i32 + padding, then pointers
    assert sharded.MsComm.user.offset == 16 and ctypes.sizeof(sharded.MsComm) == 16 + 8 * (len(header) - 3)


def test_kernel_names_available_without_gpu(pkg):
    L = pkg.lib()
    n = L.ms_kernel_count()
    names = [L.ms_kernel_name(i).decode() for i in range(n)]
    assert "ntt12_dif" in names and "leaf_hash" in names and len(set(names)) == n


def test_no_device_fails_loudly(pkg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.MstarkError):
        pkg.Context(0)


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "multi-stark_amd")):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/" not in src and "libms_oracle" not in src and "import oracle" not in src, os.path.join(dirpath, f)


def test_missing_librccl_is_an_error_code_not_a_crash(pkg, tmp_path):
    """MSAMD_RCCL_LIB names the library to load; when it cannot be loaded the call returns MS_ERR with the loader's reason
    (ms_comm_rccl_unique_id needs no GPU). Run in a child process: the library caches the RCCL it finds."""
    import subprocess
    import sys

    code = (
        "import ctypes, os, sys\n"
        "os.environ['MSAMD_RCCL_LIB'] = %r\n"
        "L = ctypes.CDLL(%r)\n"
        "L.ms_last_error.restype = ctypes.c_char_p\n"
        "buf = (ctypes.c_uint8 * 128)()\n"
        "rc = L.ms_comm_rccl_unique_id(buf)\n"
        "print(rc, L.ms_last_error().decode())\n"
        "out = ctypes.c_void_p()\n"
        "rc2 = L.ms_comm_rccl_create(None, buf, 0, 2, ctypes.byref(out))\n"
        "print(rc2)\n"
    ) % (str(tmp_path / "no_such_librccl.so"), pkg.LIB_PATH)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    first, second = r.stdout.strip().splitlines()[:2]
    assert first.startswith("-1 ") and "cannot load librccl" in first and "no_such_librccl" in first, r.stdout
    assert second.strip() == "-1"
