"""CPU: the C-ABI shared library builds, loads, and exports every symbol include/dedalus_hip.h declares."""
import os
import re

from dedalus_amd import build, libhip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    build.build_library()
    lib = libhip.load()
    header = open(os.path.join(ROOT, "include", "dedalus_hip.h")).read()
    declared = set(re.findall(r"\b(ddh_[a-z0-9_]+)\s*\(", header))
    declared -= {"ddh_handle"}
    assert len(declared) > 20
    for name in sorted(declared):
        assert hasattr(lib, name), "symbol %s declared in the header but not exported" % name
    bound = set(libhip.SIGNATURES) | {"ddh_last_error"}
    assert declared == bound, (declared ^ bound)


def test_error_reporting_without_gpu():
    lib = libhip.load()
    # invalid handle -> negative status and a message; no device work involved
    assert lib.ddh_destroy(123456789) != 0
    assert b"invalid handle" in lib.ddh_last_error()
