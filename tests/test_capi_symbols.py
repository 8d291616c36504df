"""
CPU-side checks of the boundary: the C-ABI shared library loads next to torch's HIP runtime
and exports every entry point include/prograph_hip.h declares.  No compute calls (no GPU here).
"""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared():
    text = open(os.path.join(REPO, "include", "prograph_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    from prograph_amd import _native
    declared = _declared()
    assert declared, "header parse failed"
    assert sorted(_native.SYMBOLS) == declared
    lib = _native.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pg_version() == _native.ABI_VERSION
    assert lib.pg_npad(1) == 256 and lib.pg_npad(256) == 256 and lib.pg_npad(257) == 512
    assert lib.pg_ngroups(1) == 1 and lib.pg_ngroups(32) == 1 and lib.pg_ngroups(33) == 2 and lib.pg_ngroups(128) == 4
    assert lib.pg_nchunks(64, 5) == 3 and lib.pg_nchunks(32, 5) == 2 and lib.pg_nchunks(128, 8) == 8
    assert lib.pg_scan_scratch_bytes(1) >= 16
    assert _native.npad(50_000) == lib.pg_npad(50_000) and _native.nchunks(64, 5) == lib.pg_nchunks(64, 5)


def test_one_hip_runtime_in_process():
    """The library must bind to the HIP runtime torch already mapped (same SONAME), not a second copy."""
    from prograph_amd import _native
    _native.lib()
    maps = open("/proc/self/maps").read()
    paths = set(re.findall(r"(/\S*libamdhip64\.so[^\s]*)", maps))
    assert len(paths) == 1, paths


def test_argument_validation_without_gpu():
    """Library-side argument checks return PG_E_* before any launch."""
    from prograph_amd import _native
    lib = _native.lib()
    rc = lib.pg_knn_hamming(None, 256, 0, 1, None, 256, 1, 16, 5, 4, None, None, None, None)
    assert rc == -1 and b"bad argument" in lib.pg_last_error()
    rc = lib.pg_pack_planes(ctypes.c_void_p(16), 1, 4, 300, 300, None, 5, ctypes.c_void_p(16), 256, ctypes.c_void_p(16), None)
    assert rc == -2


def test_product_fails_loudly_without_gpu(monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from prograph_amd import _native
    from prograph_amd.distance import hamming
    with pytest.raises(_native.NativeUnavailable):
        hamming(torch.tensor([[1, 2, 3]]), torch.tensor([[1, 2, 4]]))
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libprograph_hip.so")
    monkeypatch.setattr(_native, "_lib", None)
    with pytest.raises(_native.NativeUnavailable):
        _native.lib()


def test_no_product_import_of_the_oracle():
    """oracle/ is test infrastructure: nothing under prograph_amd/ may reference it."""
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "prograph_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "prograph_oracle" in src:
                    bad.append(f)
    assert not bad, bad


def test_signature_helpers_on_the_host():
    """tests/capi/sig_check.cpp: the arithmetic behind the MFMA engine's FP4 signature filter
    (prograph_amd/csrc/pg_common.h) compiled for the host - bit spreading, the ten-element encoding of a
    row's bias for every value (never above the bias, exact except -59), XOR-linearity and the
    lower-bound property of the 54-bit signature.  Runs on the CPU."""
    import subprocess
    capi = os.path.join(REPO, "tests", "capi")
    subprocess.check_call(["make", "-s", "-C", capi, "_build/sig_check"])
    out = subprocess.run([os.path.join(capi, "_build", "sig_check")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "signature helpers OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_fragment_ring_rule_holds_in_the_generated_isa():
    """pg_mm.h loads its column fragments with inline assembly the compiler cannot see through; its RULE - nothing
    touches a ring register between such a load and the wait that covers it - is checked on the generated ISA
    (tools/check_ring_asm.py; here for the cfg3 group count, every MFMA-engine instance of it: eps, symmetric eps,
    kNN with 64-lane lists, with short lists, with two row blocks; 5 and 8 bit planes)."""
    import subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "..", "tools", "check_ring_asm.py"), "2"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "RULE violations: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
