"""The C-ABI library loads on a CPU-only box, exports EVERY symbol include/ovslam_hip.h declares, and fails loudly (no CPU
fallback) when no HIP device is usable. No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ovslam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ovs_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from openvslam_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "libovslam_hip.so does not export %s" % n
    # and the ctypes table the host layer uses covers exactly the header
    assert sorted(_lib.SYMBOLS) == names


def test_fails_loudly_without_device():
    from openvslam_amd import _lib
    L = _lib.lib()
    if L.ovs_device_count() > 0:
        pytest.skip("a HIP device is visible")
    h = C.c_void_p()
    p = _lib.OrbParams(2000, 1.2, 8, 20, 7)
    assert L.ovs_orb_create(C.byref(p), 480, 752, 1, 0, C.byref(h)) == -2   # OVS_ERR_NO_DEVICE
    assert not h
    m = C.c_void_p()
    assert L.ovs_matcher_create(100, 100, 1, 0, C.byref(m)) == -2
    from openvslam_amd import feature
    with pytest.raises(_lib.OvsError):
        feature.orb_extractor()


def test_argument_validation_without_device():
    from openvslam_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    bad = _lib.OrbParams(2000, 1.0, 8, 20, 7)      # scale_factor must be > 1
    assert L.ovs_orb_create(C.byref(bad), 480, 752, 1, 0, C.byref(h)) == -1
    bad = _lib.OrbParams(2000, 1.2, 99, 20, 7)     # too many levels
    assert L.ovs_orb_create(C.byref(bad), 480, 752, 1, 0, C.byref(h)) == -1
    assert L.ovs_matcher_create(0, 10, 1, 0, C.byref(h)) == -1
    assert L.ovs_matcher_create(70000, 10, 1, 0, C.byref(h)) == -1


def test_product_does_not_import_oracle():
    """The product path must never route through oracle/ (parity claims would be void)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "openvslam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp", ".inc")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt and "ovo_" not in txt, f
