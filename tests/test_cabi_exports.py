"""CPU: the C-ABI library loads and exports every symbol include/fvhip.h declares
(no compute calls without a GPU), and fails loudly without a device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "fvhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fv_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(fv):
    import ctypes

    from fvamd import _lib

    lib = ctypes.CDLL(_lib.LIBPATH)
    declared = _declared_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), "libfvhip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "python binding lacks %s" % name
    assert set(_lib.SIGNATURES) == set(declared)
    assert fv.load().fv_abi_version() == 1


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "finitevolume.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")):
                src = open(os.path.join(dirpath, f)).read()
                assert "fv_oracle" not in src and "libfvoracle" not in src, f


def test_no_device_fails_loudly(fv):
    import ctypes

    n = ctypes.c_int(0)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        have = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        have = False
    if have:
        pytest.skip("a GPU is visible here")
    with pytest.raises(fv.FVError, match="no CPU fallback"):
        fv.Context(0)


def test_julia_shim_calls_only_declared_symbols_with_matching_arity():
    """The Julia binding was written without a Julia runtime: at least every `ccall((:name, libfvhip), ...)` must name a
    symbol of include/fvhip.h, with as many argument types as the C declaration has parameters."""
    hdr = open(os.path.join(ROOT, "include", "fvhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    arity = {}
    for name, params in re.findall(r"\b(fv_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        params = params.strip()
        arity[name] = 0 if params in ("", "void") else params.count(",") + 1
    src = open(os.path.join(ROOT, "finitevolume.jl_amd", "julia", "FiniteVolumeHIP.jl")).read()
    calls = re.findall(r"ccall\(\(:(\w+),\s*libfvhip\),\s*\w+(?:\{[^}]*\})?,\s*\(([^)]*)\)", src)
    assert len(calls) >= 20
    for name, types in calls:
        assert name in arity, "the shim calls %s, which include/fvhip.h does not declare" % name
        ntypes = len([t for t in re.split(r",(?![^{]*\})", types) if t.strip()])
        assert ntypes == arity[name], "%s: %d argument types in the shim, %d parameters in the header" % (name, ntypes, arity[name])
