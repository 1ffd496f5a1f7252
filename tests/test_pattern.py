"""Both copies of the rBRIEF pattern (oracle side, product side) are the same 256x4 table with the recorded hash."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHA = "3f202c09967ef499081baca510075490fe60c9626ed8e748ff8e94378229a598"


def _read(path):
    txt = open(os.path.join(ROOT, path)).read()
    txt = "\n".join(l for l in txt.splitlines() if not l.strip().startswith("//"))
    return [int(v) for v in re.findall(r"-?\d+", txt)]


def test_pattern_copies():
    a = _read("oracle/orb_pattern.inc")
    b = _read("openvslam_amd/csrc/orb_pattern.inc")
    assert a == b and len(a) == 1024
    assert hashlib.sha256(bytes(v + 128 for v in a)).hexdigest() == SHA
    assert min(a) == -13 and max(a) == 12
    # the rows every reprint of OpenCV's bit_pattern_31_ (orb.cpp; ORB-SLAM2's ORBextractor.cc) opens and closes with
    assert a[:20] == [8, -3, 9, 5, 4, 2, 7, -12, -11, 9, -8, 2, 7, -12, 12, -13, 2, -13, 2, 12]
    assert a[-8:] == [7, 0, 12, -2, -1, -6, 0, -11]
    # every sample stays inside the radius the extractor reserves (orb_patch_radius_ = 19 after rotation + rounding)
    import math
    assert max(math.hypot(a[i], a[i + 1]) for i in range(0, 1024, 2)) < 18.5


def test_oracle_exports_same_table(oracle):
    import numpy as np
    a = np.array(_read("oracle/orb_pattern.inc")).reshape(256, 4)
    assert np.array_equal(oracle.orb_pattern(), a)
