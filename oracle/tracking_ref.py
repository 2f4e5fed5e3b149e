"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Independent CPU restatements for the tracking glue ("next" rows N1/N2):
  crop_affine_ref      the crop geometry of lib/pose/utils/transforms.py:173-184,231-240 (cv2.warpAffine with
                       t = get_transform(center, scale, res): inverse map, bilinear, constant-0 border), in
                       float64 numpy: the IDEAL bilinear crop (what `ft_crop_affine_fwd` computes in fp32).
  warp_affine_cv2_ref  what the reference's `cv2.warpAffine(img_u8, t[:2], (w, h))` RETURNS for a uint8 frame: OpenCV's
                       fixed-point INTER_LINEAR (round 5).  cv2 is a third-party dependency that is absent from this image
                       and that the reference does not pin (no requirements file); restated from the published algorithm of
                       OpenCV 2.4 ... 4.10, modules/imgproc/src/imgwarp.cpp: `cv::warpAffine` (matrix inversion in double),
                       `WarpAffineInvoker::operator()` (AB_BITS = 10 fixed-point inverse map, round_delta, 1/32-px
                       fractional index), `initInterTab2D` (the 32 x 32 table of 15-bit bilinear weights),
                       `remapBilinear<FixedPtCast<int, uchar, 15>, ...>` ((sum + 2^14) >> 15, constant-0 border taps).
                       OpenCV >= 4.11 is reported to route INTER_LINEAR through new floating-point SIMD kernels whose
                       last bit can differ; the restatement is of the classic path.  No cv2 here => checked against
                       hand-derived known answers only (tests/test_oracle_cpu.py): parity UNPINNED, integer-exact by
                       construction.  `ft_crop_affine_cv2_fwd` must reproduce it BIT FOR BIT.
  cv2_crop_matrix      get_transform (transforms.py:173-184, rot = 0) + the inversion of `cv::warpAffine`, in float64.
  nms_ref              literal loop restatement of lib/detection/nms/src/nms.c:33-64 (IoU >= thresh, +1 widths)
                       with the score-descending order of pth_nms.py:16.
  box_propagation_ref  intended semantics of lib/tracking/flow_utils.py:7-35, per person / per joint loops
                       (the reference function itself does not run: wrong H/W axes, numpy-2-incompatible index).
"""
from __future__ import annotations

import numpy as np


def crop_affine_ref(img: np.ndarray, center, scale, res, mean=None, inv_std=None, pre_scale=1.0) -> np.ndarray:
    """img [H,W,C] uint8 -> [C,rh,rw] float32."""
    H, W, C = img.shape
    rh, rw = res
    t = np.eye(3)
    t[0, 0] = t[1, 1] = rh / scale
    t[0, 2] = -rh * center[0] / scale + 0.5 * rw
    t[1, 2] = -rh * center[1] / scale + 0.5 * rh
    tinv = np.linalg.inv(t)
    ys, xs = np.meshgrid(np.arange(rh), np.arange(rw), indexing="ij")
    src = tinv @ np.stack((xs.ravel(), ys.ravel(), np.ones(rh * rw)))
    sx, sy = src[0].reshape(rh, rw), src[1].reshape(rh, rw)
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    ax, ay = sx - x0, sy - y0
    f = img.astype(np.float64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return np.where(ok[..., None], f[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)
    out = ((1 - ay) * (1 - ax))[..., None] * tap(y0, x0) + ((1 - ay) * ax)[..., None] * tap(y0, x0 + 1) + \
          (ay * (1 - ax))[..., None] * tap(y0 + 1, x0) + (ay * ax)[..., None] * tap(y0 + 1, x0 + 1)
    out = out * pre_scale
    if mean is not None:
        out = out - np.asarray(mean)
    if inv_std is not None:
        out = out * np.asarray(inv_std)
    return out.transpose(2, 0, 1).astype(np.float32)


# ---- OpenCV's uint8 warpAffine / INTER_LINEAR, restated (see the header) ---------------------------------------------
INTER_BITS = 5                      # imgproc.hpp: INTER_BITS, INTER_TAB_SIZE = 32
INTER_TAB_SIZE = 1 << INTER_BITS
INTER_REMAP_COEF_BITS = 15          # imgproc.hpp: INTER_REMAP_COEF_SCALE = 1 << 15
AB_BITS = 10                        # imgwarp.cpp WarpAffineInvoker: AB_BITS = MAX(10, INTER_BITS)


def get_transform_ref(center, scale, res):
    """lib/pose/utils/transforms.py:173-184 with factor = 1, rot = 0, delta = 0, same float64 operation order."""
    t = np.eye(3)
    res_ = res[0] * 1
    t[0, 0] = res_ / scale
    t[1, 1] = res_ / scale
    t[0, 2] = -res_ * center[0] / scale + 0.5 * res[1] + 0
    t[1, 2] = -res_ * center[1] / scale + 0.5 * res[0] + 0
    return t


def cv2_invert_affine(M):
    """`cv::warpAffine` without WARP_INVERSE_MAP: the 2x3 matrix is inverted in double, in this operation order
    (imgwarp.cpp, cv::warpAffine: `double D = M[0]*M[4] - M[1]*M[3]; D = D != 0 ? 1./D : 0; ...`)."""
    M = [float(v) for v in np.asarray(M, dtype=np.float64).reshape(6)]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11 = M[4] * D
    A22 = M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2] = b1
    M[5] = b2
    return np.asarray(M, dtype=np.float64)


def cv2_crop_matrix(center, scale, res):
    """The dst -> src map cv2 uses for the reference's crop: invert(get_transform(...)[:2]) -> float64[6]."""
    return cv2_invert_affine(get_transform_ref(center, scale, res)[:2])


def cv2_bilinear_tab():
    """`initInterTab2D(INTER_LINEAR, fixpt = true)`: [1024][4] 15-bit integer weights (w00, w01, w10, w11) for the
    fractional index fy * 32 + fx.  Every weight is (a / 32)(b / 32) * 2^15 = a * b * 32, exact; the only entry whose
    rounding matters is (0, 0): 1.0 * 2^15 saturates to short 32767, the table's sum-to-2^15 fix-up then adds the
    missing 1 to the largest weight it scans, which for the 2 x 2 kernel is the w11 slot (its scan starts at
    [ksize/2][ksize/2] = [1][1]) -> (32767, 0, 0, 1).  For uint8 pixels both forms of that entry give the same
    result (checked exhaustively in tests/test_oracle_cpu.py), so nothing below depends on this reading."""
    tab = np.zeros((INTER_TAB_SIZE * INTER_TAB_SIZE, 4), dtype=np.int64)
    for fy in range(INTER_TAB_SIZE):
        for fx in range(INTER_TAB_SIZE):
            tab[fy * INTER_TAB_SIZE + fx] = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    tab[0] = [32767, 0, 0, 1]
    return tab


def _cv_round(v):
    """cvRound(double) = lrint: round half to even; out-of-range values give INT_MIN like cvtsd2si."""
    r = np.rint(np.asarray(v, dtype=np.float64))
    bad = ~(np.abs(r) < 2147483648.0)
    return np.where(bad, -2147483648, np.where(bad, 0, r)).astype(np.int64)


def _wrap32(v):
    return ((np.asarray(v, dtype=np.int64) + 2147483648) & 0xFFFFFFFF) - 2147483648


def warp_affine_cv2_ref(img: np.ndarray, Minv, dsize_hw, tab=None) -> np.ndarray:
    """img [H,W,C] uint8, Minv = float64[6] dst -> src map (cv2_invert_affine of the matrix handed to cv2.warpAffine)
    -> [h, w, C] uint8, INTER_LINEAR, BORDER_CONSTANT 0."""
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, C = img.shape
    h, w = dsize_hw
    M = np.asarray(Minv, dtype=np.float64).reshape(6)
    AB_SCALE = float(1 << AB_BITS)
    round_delta = (1 << AB_BITS) // INTER_TAB_SIZE // 2                  # 16
    xs = np.arange(w, dtype=np.float64)
    ys = np.arange(h, dtype=np.float64)
    adelta = _cv_round(M[0] * xs * AB_SCALE)                              # WarpAffineInvoker: adelta[x], bdelta[x]
    bdelta = _cv_round(M[3] * xs * AB_SCALE)
    X0 = _wrap32(_cv_round((M[1] * ys + M[2]) * AB_SCALE) + round_delta)  # per row
    Y0 = _wrap32(_cv_round((M[4] * ys + M[5]) * AB_SCALE) + round_delta)
    X = _wrap32(X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = _wrap32(Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)                          # saturate_cast<short>
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    fxy = (Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))
    wt = (cv2_bilinear_tab() if tab is None else tab)[fxy]               # [h, w, 4]
    f = img.astype(np.int64)

    def tap(yy, xx):                                                      # remapBilinear, BORDER_CONSTANT: cval = 0
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return np.where(ok[..., None], f[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0)
    acc = tap(sy, sx) * wt[..., 0:1] + tap(sy, sx + 1) * wt[..., 1:2] + tap(sy + 1, sx) * wt[..., 2:3] + tap(sy + 1, sx + 1) * wt[..., 3:4]
    out = (acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS  # FixedPtCast<int, uchar, 15>
    return np.clip(out, 0, 255).astype(np.uint8)


def crop_cv2_ref(img: np.ndarray, center, scale, res) -> np.ndarray:
    """`transform_image(img, center, scale, res)` of the reference for a uint8 HWC frame -> [C, h, w] uint8
    (transforms.py:231-240: warpAffine, then HWC -> CHW)."""
    return warp_affine_cv2_ref(img, cv2_crop_matrix(center, scale, res), res).transpose(2, 0, 1)


def nms_ref(dets: np.ndarray, thresh: float):
    dets = np.asarray(dets, dtype=np.float32)
    n = dets.shape[0]
    areas = (dets[:, 2] - dets[:, 0] + 1) * (dets[:, 3] - dets[:, 1] + 1)
    order = sorted(range(n), key=lambda i: -dets[i, 4])
    suppressed = [False] * n
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        for _j in range(_i + 1, n):
            j = order[_j]
            if suppressed[j]:
                continue
            w = max(0.0, min(dets[i, 2], dets[j, 2]) - max(dets[i, 0], dets[j, 0]) + 1)
            h = max(0.0, min(dets[i, 3], dets[j, 3]) - max(dets[i, 1], dets[j, 1]) + 1)
            inter = w * h
            if inter / (areas[i] + areas[j] - inter) >= thresh:
                suppressed[j] = True
    return keep


def box_propagation_ref(keypoints: np.ndarray, flow: np.ndarray, extend_factor=0.15) -> np.ndarray:
    _, H, W = flow.shape
    boxes = []
    for person in keypoints:
        pts = []
        for (x, y, s) in person:
            xi, yi = min(max(int(x), 0), W - 1), min(max(int(y), 0), H - 1)
            pts.append((x + flow[0, yi, xi], y + flow[1, yi, xi], s))
        vis = [(px, py) for px, py, s in pts if s > 0]
        mn = np.array([min([p[0] for p in vis] + ([] if len(vis) == len(pts) else [max(H, W)])),
                       min([p[1] for p in vis] + ([] if len(vis) == len(pts) else [max(H, W)]))]) if vis else np.array([max(H, W)] * 2, float)
        mx = np.array([max([p[0] for p in vis] + ([] if len(vis) == len(pts) else [0.0])),
                       max([p[1] for p in vis] + ([] if len(vis) == len(pts) else [0.0]))]) if vis else np.zeros(2)
        ext = (mx - mn) * extend_factor / 2
        ul = np.maximum(mn - ext, 0)
        br = np.minimum(mx + ext, [W - 1, H - 1])
        boxes.append(np.concatenate((ul, br)))
    return np.asarray(boxes)
