"""CPU: the F(4x4, 3x3) matrices csrc/winograd4.hip hard-codes (B^T, G, A^T: interpolation points 0, 1,
-1, 2, -1/2, inf) reproduce a 3x3 correlation exactly in float64, tile by tile, for dilation 1 and through
the (y % d, x % d) sub-grids for dilation 2; and the fp32 emulation of tests/wino_error_study.py
(the tool that sized the form's rounding before it was built) stays within 2e-5 of max|y| on white
inputs - the bound tests/test_gpu_h2.py::test_conv3x3_winograd4_f16x2 holds the kernel to."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

BT = np.array([[2, 3, -4, -3, 2, 0], [0, 2, 5, 1, -2, 0], [0, 2, 1, -5, 2, 0],
               [0, -1, -2, 1, 2, 0], [0, -2, 1, 2, -1, 0], [0, 2, 3, -4, -3, 2]], np.float64)
G = np.array([[1 / 2, 0, 0], [1 / 6, 1 / 6, 1 / 6], [1 / 6, -1 / 6, 1 / 6],
              [1 / 30, 1 / 15, 2 / 15], [16 / 15, -8 / 15, 4 / 15], [0, 0, 1 / 2]], np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -1 / 2, 0], [0, 1, 1, 4, 1 / 4, 0],
               [0, 1, -1, 8, -1 / 8, 1]], np.float64)


def test_one_tile_equals_the_correlation():
    rng = np.random.default_rng(3)
    d = rng.standard_normal((6, 6))
    g = rng.standard_normal((3, 3))
    y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
    want = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(4)] for i in range(4)])
    np.testing.assert_allclose(y, want, rtol=0, atol=1e-12)
    # the bounds the kernel's operand scale relies on: |B^T d B| <= 196 max|d|
    assert np.abs(BT).sum(axis=1).max() == 14.0


def test_study_tool_emulation_matches_conv2d():
    import wino_error_study as ws
    for m in (2, 4, 5):
        bt, g, at = ws.mats(m)
        if m == 5:
            assert np.array_equal(bt, BT) and np.allclose(g, G) and np.array_equal(at, AT)
    rng = np.random.default_rng(4)
    x = np.maximum(rng.standard_normal((32, 21, 30)), 0).astype(np.float32)
    w = (rng.standard_normal((48, 32, 3, 3)) * np.sqrt(2.0 / (9 * 32))).astype(np.float32)
    b = rng.standard_normal(48).astype(np.float32)
    for dil in (1, 2):
        ref = F.conv2d(torch.from_numpy(x)[None].double(), torch.from_numpy(w).double(),
                       torch.from_numpy(b).double(), padding=dil, dilation=dil)[0].numpy()
        for m, split in ((2, False), (4, False), (5, False), (5, True)):
            y = ws.wino_layer(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(b), dil, m,
                              split)[0].numpy()
            err = np.abs(y - ref).max() / np.abs(ref).max()
            assert err < (2e-6 if m == 2 else 2e-5), (dil, m, split, err)
