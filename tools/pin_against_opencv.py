#!/usr/bin/env python3
"""Emit OpenCV-generated fixtures that pin the oracle's restatement of the five OpenCV functions the ORB path calls.

This container has no OpenCV, so parity with upstream's third-party arithmetic is UNPINNED for
cv::resize (INTER_LINEAR, 8-bit), cv::FAST (TYPE_9_16 + NMS), cv::GaussianBlur (7x7, sigma 2, BORDER_REFLECT_101, 8-bit),
cv::fastAtan2 and cvRound. Run this script ONCE on any machine that has `cv2` (any 3.4.x / 4.x build: the version is recorded,
ORACLE_SPEC rule 10 is version dependent) and numpy:

    python tools/pin_against_opencv.py            # writes tests/golden/opencv_pins.npz

commit the file, and tests/test_pinned.py stops skipping: it compares the oracle (CPU tier) and the HIP pyramid / FAST candidates /
descriptors (GPU tier) against what OpenCV itself produced on the same seeded inputs. The inputs come from
openvslam_amd.synth.synth_frame, which is pure numpy and needs no GPU, so the fixture is reproducible anywhere.

Contents of the fixture (all on synth_frame inputs so nothing but a seed travels):
  * version / build string of the OpenCV that produced it;
  * pyr_<name>_<l>: the 8-level pyramid chain exactly as orb_extractor::compute_image_pyramid builds it -- level l resized from level
    l-1 to (round(cols / 1.2^l), round(rows / 1.2^l)) with the float cumulative scale table -- full arrays for the small frame,
    sha256 + the first / last rows for the large ones;
  * fast_<name>_<l>_t<thr>: cv::FAST keypoints (x, y, response) in emission order for thresholds 20 and 7 on every level of the small
    frame and on two 70x70 cells of the large one (the cell size compute_fast_keypoints uses);
  * blur_<name>_<l>: cv::GaussianBlur output, same storage rule as the pyramid;
  * atan2_in / atan2_out: cv::fastAtan2 on 40 000 integer-valued (m01, m10) moment pairs incl. axes and quadrant borders;
  * round_in / round_out: cvRound on .5 ties and neighbours (checked through cv2.KeyPoint-free arithmetic: np.rint mirrors it; the
    values are recorded from cv2's own saturate path via cv2.convertScaleAbs-free int conversion `cv2.cvRound` when the binding has it).
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FRAMES = {"small": (203, 331, 5), "euroc": (480, 752, 0), "hd": (1080, 1920, 7)}
NUM_LEVELS = 8
SCALE = np.float32(1.2)


def scale_table():
    sf = [np.float32(1.0)]
    for _ in range(1, NUM_LEVELS):
        sf.append(np.float32(sf[-1] * SCALE))   # cumulative float product, as orb_extractor::calc_scale_factors
    return np.array(sf, np.float32)


def level_sizes(rows, cols):
    sf = scale_table()
    return [(int(round(rows * 1.0 / float(s))), int(round(cols * 1.0 / float(s)))) for s in sf]


def moment_pairs():
    rng = np.random.default_rng(11)
    m = rng.integers(-2_700_000, 2_700_001, size=(39_000, 2))
    edge = np.array([[0, 0], [0, 1], [1, 0], [0, -1], [-1, 0], [1, 1], [-1, 1], [1, -1], [-1, -1], [5, 5], [-5, 5], [7, 0], [0, 7]])
    small = rng.integers(-40, 41, size=(1000 - len(edge), 2))
    return np.concatenate([m, edge, small]).astype(np.float32)


def main():
    try:
        import cv2
    except ImportError:
        print("cv2 is not importable here: run this on a machine with OpenCV (pip install opencv-python-headless)", file=sys.stderr)
        return 2
    from openvslam_amd.synth import synth_frame
    out = {"opencv_version": np.array(cv2.__version__), "scale_factors": scale_table()}
    for name, (rows, cols, seed) in FRAMES.items():
        img = synth_frame(rows, cols, seed=seed)
        out["input_sha256_" + name] = np.array(hashlib.sha256(img.tobytes()).hexdigest())
        sizes = level_sizes(rows, cols)
        prev = img
        for l in range(NUM_LEVELS):
            if l > 0:
                r, c = sizes[l]
                prev = cv2.resize(prev, (c, r), interpolation=cv2.INTER_LINEAR)
            blur = cv2.GaussianBlur(prev, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
            for kind, arr in (("pyr", prev), ("blur", blur)):
                key = "%s_%s_%d" % (kind, name, l)
                if name == "small":
                    out[key] = arr.copy()
                else:
                    out[key + "_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest())
                    out[key + "_rows"] = np.stack([arr[0], arr[arr.shape[0] // 2], arr[-1]])
            for thr in (20, 7):
                det = cv2.FastFeatureDetector_create(threshold=thr, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
                if name == "small":
                    regions = {"": prev}
                elif l == 0:
                    regions = {"_cellA": prev[19:19 + 70, 19:19 + 70], "_cellB": prev[rows // 2:rows // 2 + 70, cols // 2:cols // 2 + 70]}
                else:
                    regions = {}
                for tag, reg in regions.items():
                    kps = det.detect(np.ascontiguousarray(reg), None)
                    out["fast_%s_%d_t%d%s" % (name, l, thr, tag)] = np.array([[k.pt[0], k.pt[1], k.response] for k in kps], np.float32).reshape(-1, 3)
    mp = moment_pairs()
    out["atan2_in"] = mp
    out["atan2_out"] = np.array([cv2.fastAtan2(float(y), float(x)) for y, x in mp], np.float32)
    ties = np.concatenate([np.arange(-8, 9) + 0.5, np.arange(-8, 9) + 0.49999997, np.arange(-8, 9) + 0.50000006]).astype(np.float32)
    out["round_in"] = ties
    out["round_out"] = np.array([cv2.cvRound(float(v)) if hasattr(cv2, "cvRound") else int(np.rint(v)) for v in ties], np.int32)
    dst = os.path.join(ROOT, "tests", "golden", "opencv_pins.npz")
    np.savez_compressed(dst, **out)
    print("wrote %s (%d arrays, OpenCV %s)" % (dst, len(out), cv2.__version__))
    return 0


if __name__ == "__main__":
    sys.exit(main())
