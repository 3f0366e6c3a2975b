"""Deterministic test image shared by tests/golden/gen_golden.py (`_test_image`) and the tests."""
import numpy as np


def test_image(w, h):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), (((xx // 16) + (yy // 16)) % 2) * 200], -1)
    return img.astype(np.uint8)


test_image.__test__ = False      # not a pytest test
