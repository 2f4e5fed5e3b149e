"""Impulse-response known-answer tests for the Correlation operator, derived BY HAND from the index
arithmetic of lib/flownet/networks/correlation_package/src/correlation_cuda_kernel.cu:34-106 — no reference code runs
(its CPU entry points are empty stubs, correlation.c:3-33), so this does not lift the "parity unpinned" cap; it pins the
things a restatement can silently get wrong: which axis tj / ti displace, the sign of the displacement, the channel
order tc = (tj + r) * D + (ti + r), the pad < max_displacement output window, the k x k window and stride1.

Derivation.  rInput* are the inputs zero-padded by `pad` (kernel :10-32).  Output pixel (y, x) reads rInput1 around
(y1, x1) = (y*s1 + d + krad, x*s1 + d + krad) and rInput2 around (y1 + tj*s2, x1 + ti*s2) (:53-54, :76-77), sums the
products over the k x k window and all channels and divides by nelems = k*k*C (:65, :99).  With ONE non-zero element in
each input, in1[c0, ya, xa] = alpha and in2[c0, yb, xb] = beta, the only non-zero outputs are
    tj = (yb - ya) / s2, ti = (xb - xa) / s2            (must divide, |tj|, |ti| <= r = d / s2)
    channel tc = (tj + r) * D + (ti + r),  D = 2r + 1   (:97)
    pixels  y = (ya + pad - d - krad - j) / s1,  x = (xa + pad - d - krad - i) / s1  for window offsets j, i in [-krad, krad]
    value   alpha * beta / (k*k*C)
Run once: python tests/golden/make_correlation_kat.py -> tests/golden/correlation_kat.npz (inputs + expected outputs).
"""
import os

import numpy as np

CASES = [
    # name, C, H, W, pad, k, d, s1, s2, impulse1 (c, y, x, alpha), impulse2 (c, y, x, beta)
    ("rows_are_tj_cols_are_ti", 3, 5, 7, 4, 1, 4, 1, 2, (1, 2, 3, 2.0), (1, 4, 1, 3.0)),
    ("pad_smaller_than_max_disp", 2, 6, 9, 2, 1, 4, 1, 2, (0, 3, 4, 1.5), (0, 3, 8, 2.0)),
    ("window_3x3", 2, 6, 6, 3, 3, 2, 1, 1, (1, 2, 3, 1.0), (1, 3, 1, 4.0)),
    ("stride1_2", 1, 8, 8, 2, 1, 2, 2, 1, (0, 4, 6, 2.0), (0, 5, 4, 0.5)),
    ("flownetc_shape_small", 4, 6, 8, 20, 1, 20, 1, 2, (2, 1, 5, 1.0), (2, 5, 1, 8.0)),
]


def expected(C, H, W, pad, k, d, s1, s2, imp1, imp2):
    krad, r = (k - 1) // 2, d // s2
    D = 2 * r + 1
    border = d + krad
    oh = -(-(H + 2 * pad - 2 * border) // s1)
    ow = -(-(W + 2 * pad - 2 * border) // s1)
    out = np.zeros((1, D * D, oh, ow), np.float32)
    (c1, ya, xa, alpha), (c2, yb, xb, beta) = imp1, imp2
    assert c1 == c2 and (yb - ya) % s2 == 0 and (xb - xa) % s2 == 0
    tj, ti = (yb - ya) // s2, (xb - xa) // s2
    assert abs(tj) <= r and abs(ti) <= r
    tc = (tj + r) * D + (ti + r)
    for j in range(-krad, krad + 1):
        for i in range(-krad, krad + 1):
            ny, nx = ya + pad - d - krad - j, xa + pad - d - krad - i
            if ny % s1 == 0 and nx % s1 == 0 and 0 <= ny // s1 < oh and 0 <= nx // s1 < ow:
                out[0, tc, ny // s1, nx // s1] += alpha * beta / (k * k * C)
    return out


def main():
    blob = {}
    for name, C, H, W, pad, k, d, s1, s2, imp1, imp2 in CASES:
        a = np.zeros((1, C, H, W), np.float32)
        b = np.zeros((1, C, H, W), np.float32)
        a[0, imp1[0], imp1[1], imp1[2]] = imp1[3]
        b[0, imp2[0], imp2[1], imp2[2]] = imp2[3]
        blob[name + ".in1"], blob[name + ".in2"] = a, b
        blob[name + ".params"] = np.array([pad, k, d, s1, s2], np.int32)
        blob[name + ".out"] = expected(C, H, W, pad, k, d, s1, s2, imp1, imp2)
        assert blob[name + ".out"].any(), name
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "correlation_kat.npz"), **blob)


if __name__ == "__main__":
    main()
