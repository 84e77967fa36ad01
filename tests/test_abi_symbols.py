"""CPU-side checks of the C-ABI: the library builds, loads, and exports every symbol include/*.h declares.
No compute calls (there is no GPU here); the one call made must fail loudly with LSA_ERR_NO_DEVICE."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from lattisense_amd import build, _native
    build.build_native()
    return _native


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b((?:lsa|create|release|bind|run)_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_every_declared_symbol_is_exported(native):
    L = native.lib()
    for header in ("lattisense_amd.h", "lattisense_task.h"):
        if not os.path.exists(os.path.join(ROOT, "include", header)):
            continue
        names = _declared(header)
        assert names, header
        for n in names:
            assert hasattr(L, n), "%s declares %s but the .so does not export it" % (header, n)


def test_binding_table_matches_header(native):
    declared = set(_declared("lattisense_amd.h"))
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)


def test_no_device_fails_loudly(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lattisense_amd import params
    from lattisense_amd.device import DeviceContext
    P = params.BFV_DEFAULT[8192]
    with pytest.raises(native.LsaError) as e:
        DeviceContext(0, 8192, P["q"], P["p"], P["t"])
    assert e.value.code == 2  # LSA_ERR_NO_DEVICE: no CPU fallback exists
