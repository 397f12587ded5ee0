"""The C-ABI library loads and exports every symbol include/rt_hip.h declares
(no compute calls: this runs without a GPU)."""
import os
import re

from raytracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), "librt_hip.so does not export %s" % name
        assert name in capi.SYMBOLS, "capi.py does not bind %s" % name
    assert sorted(capi.SYMBOLS) == names


def test_error_reporting_without_a_context():
    lib = capi.load()
    assert lib.rt_set_max_bounces(None, 6) != 0
    assert b"null context" in lib.rt_last_error()


def test_creation_fails_loudly_without_a_gpu():
    lib = capi.load()
    if lib.rt_device_count() > 0:
        return
    try:
        capi.Context(16, 16)
    except capi.RtError as e:
        assert "rt_hip error" in str(e)
    else:
        raise AssertionError("rt_create succeeded without a GPU")
