"""CPU-side checks of the drop-in boundary: libpvlm.so loads and exports every symbol
include/pvlm.h declares (no compute calls without a GPU), and fails loudly without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "pvlm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pvlm_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    from panovlm_amd import api, build
    lib = ctypes.CDLL(build.build())
    declared = header_symbols()
    assert len(declared) >= 30
    for s in declared:
        assert hasattr(lib, s), "libpvlm.so does not export " + s
    assert sorted(api.ABI_SYMBOLS) == declared, set(api.ABI_SYMBOLS) ^ set(declared)


def test_no_torch_types_and_c_linkage():
    txt = open(os.path.join(ROOT, "include", "pvlm.h")).read()
    assert 'extern "C"' in txt and "torch" not in txt and "at::" not in txt


def test_product_never_touches_the_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "panovlm_amd")):
        if os.sep + "build" in d:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                s = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|liboracle|oracle/", s):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_context_creation_fails_loudly_without_gpu():
    import panovlm_amd as pv
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = os.path.exists("/dev/kfd")
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(pv.PvlmError):
        pv.Context(0)


def test_header_is_plain_c_and_cxx():
    """include/pvlm.h is the drop-in boundary: it must compile as C99 (cgo / JNI / ctypes-style bindings) and as C++11,
    warning-free, with nothing but the standard headers."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "pvlm.h")
    for cmd in (["gcc", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", hdr],
                ["g++", "-x", "c++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", hdr]):
        r = subprocess.run(cmd, capture_output=True)
        assert r.returncode == 0, r.stderr.decode()
