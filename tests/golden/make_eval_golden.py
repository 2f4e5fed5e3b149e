"""Golden vectors for the remaining host-side evaluation helpers of lib/pose/utils/evaluation.py (nms_heatmap :37-59,
calc_dists / dist_acc / accuracy :104-159, compute_pck :164-175), produced by IMPORTING the reference in the build
container (it cannot travel).  Run: python tests/golden/make_eval_golden.py  ->  tests/golden/eval_extra_golden.npz

torch drift: under torch 2.x `indexes.div(w)` is a true division, so the reference's y comes out as y + x / W (the relation
make_golden.py asserts for max_preds).  It is asserted again here and the torch-0.4 values (floor) are what is stored; the
`accuracy` case keeps every predicted peak in the column of its target peak, where the drift cancels in the distance.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden import import_reference_pose  # noqa: E402


def main():
    _, ref_eval = import_reference_pose()
    rng = np.random.RandomState(11)
    out = {}
    # ---- nms_heatmap: smooth random maps + planted plateaus (ties) ------------------------------------------------
    hm = torch.from_numpy(rng.normal(0, 1, (3, 5, 16, 12)).astype(np.float32))
    hm[0, 0, 4, 5] = hm[0, 0, 4, 6] = 9.0           # a two-pixel plateau: both survive the 3x3 test, first index wins
    hm[1, 2] = -1.0                                  # constant negative map: every pixel is a "peak", scores stay negative
    for thr, win in ((0, 3), (0.5, 3), (0, 5)):
        ref = ref_eval.nms_heatmap(hm.clone(), threshold=thr, window_size=win)
        x, y_ref = ref[..., 0], ref[..., 1]
        y = np.floor(y_ref)
        assert np.allclose(y_ref, y + x / hm.shape[3], atol=1e-5), "drift relation y_ref = y + x / W"
        ref[..., 1] = y
        out[f"nms_thr{thr}_win{win}"] = ref
    out["nms_heatmap"] = hm.numpy()
    # ---- calc_dists / dist_acc / compute_pck: plain numpy in the reference ------------------------------------------
    preds = rng.uniform(0, 40, (4, 6, 2))
    target = preds + rng.normal(0, 3.0, (4, 6, 2))
    target[1, 2] = 0.5                               # "not annotated" (< 1): distance -1
    norm = np.ones((4, 2)) * np.array([64, 48]) / 10
    dists = ref_eval.calc_dists(preds, target, norm)
    out.update(cd_preds=preds, cd_target=target, cd_norm=norm, cd_dists=dists,
               cd_acc=np.array([ref_eval.dist_acc(dists[c]) for c in range(6)]),
               cd_acc_all_skipped=np.array(ref_eval.dist_acc(np.full((5,), -1.0))))
    pred3 = np.concatenate((preds, rng.uniform(0, 1, (4, 6, 1))), axis=2)
    anno3 = np.concatenate((target, (rng.uniform(0, 1, (4, 6, 1)) > 0.3).astype(np.float64)), axis=2)
    ref_scale = rng.uniform(2, 6, (4, 1))
    out.update(pck_pred=pred3, pck_anno=anno3, pck_scale=ref_scale, pck=ref_eval.compute_pck(pred3, anno3, ref_scale, 0.5))
    # ---- accuracy (PCK on heat maps): predicted peak = target peak moved along y only -----------------------------
    n, c, h, w = 3, 4, 20, 16
    tgt = np.zeros((n, c, h, w), np.float32)
    outp = np.zeros((n, c, h, w), np.float32)
    for i in range(n):
        for j in range(c):
            ty, tx = rng.randint(2, h - 6), rng.randint(1, w)
            tgt[i, j, ty, tx] = 1.0
            outp[i, j, ty + rng.randint(0, 5), tx] = 1.0
    tgt[2, 1] = 0.0                                  # empty target map: peak at (0, 0) -> skipped (< 1)
    acc, avg, cnt, pred = ref_eval.accuracy(torch.from_numpy(outp), torch.from_numpy(tgt))
    out.update(acc_out=outp, acc_tgt=tgt, acc=acc, acc_avg=np.array(avg), acc_cnt=np.array(cnt))
    np.savez_compressed(os.path.join(HERE, "eval_extra_golden.npz"), **out)
    print("written:", sorted(out))


if __name__ == "__main__":
    main()
