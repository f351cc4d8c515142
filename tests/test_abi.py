"""CPU: the C-ABI library loads and exports exactly what include/nerfacc_hip.h declares."""
import os
import re

from conftest import ROOT


def test_header_and_library_agree():
    from nerfacc_amd import _backend as B
    hdr = open(os.path.join(ROOT, "include", "nerfacc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nfa_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = B.load()  # builds with hipcc if needed; raises if a bound symbol is missing
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in nerfacc_hip.h but not exported"
    assert declared == set(B.EXPORTED_SYMBOLS), declared ^ set(B.EXPORTED_SYMBOLS)
    assert lib.nfa_version() >= 100
    from nerfacc_amd._backend import seg_plan
    assert seg_plan(0) == (1024, 1) and seg_plan(32 * 1024 * 1024)[0] % 256 == 0
    assert seg_plan(2048, 1000) == (1024, 2 + 1000 // 256 + 1)   # a tile also ends after 256 rays


def test_argument_errors_are_reported():
    from nerfacc_amd import _backend as B
    import ctypes as C
    lib = B.load()
    rc = lib.nfa_traverse_grids(None, None)
    assert rc != 0 and b"null args" in lib.nfa_last_error()
    a = B.TraverseArgs()
    a.n_rays = 4
    a.mode = 7
    rc = lib.nfa_traverse_grids(C.byref(a), None)
    assert rc != 0 and b"mode" in lib.nfa_last_error()
    rc = lib.nfa_importance_sampling(None, None, None, 4, 3, 0, 0, 0, 0, None, None, None)
    assert rc != 0 and b">= 1" in lib.nfa_last_error()
