# -*- coding:utf-8 -*-
"""Pose entry point on the HIP path — inference subset of the reference's tools/pose/main.py.

    from tools.pose.main import main
    out = main(model='deconv', backbone='resnet50', dataset='coco', input_res=(256, 192),
               resume='deconv_resnet50_best.pth', flip_test=True)

Same protocol as the reference (main.py:42-198): kwargs -> opt.parse, `getattr(models, opt.model)(opt.backbone,
num_classes=..., pretrained=...)`, `.cuda()`, optional resume of `<checkpoint_path>/<dataset>/<exp_id>/<resume>`
(`checkpoint['state_dict']`), then `validate` (main.py:254-372): forward, optional flip test, final_preds.
Datasets, training, losses and tensorboard are out of scope: crops come from a caller-supplied
iterable of (inputs[B,3,H,W], meta) or, by default, from the synthetic generator.  The flip test runs on the
device (flip + left/right channel swap) instead of the reference's numpy round trip (main.py:289-299).
"""
from __future__ import print_function, absolute_import

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from flowtrack.pytorch_amd import synth                      # noqa: E402
from flowtrack.pytorch_amd.pose import evaluation, models    # noqa: E402
from tools.pose.config import opt                            # noqa: E402

num_joints = {'mpii': 16, 'aic': 14, 'coco': 17}
# left/right joint pairs per dataset: the (joint_right, joint_left) tables of get_pairs, lib/pose/utils/transforms.py:60-95,
# which swaplr_image applies with opt.dataset (tools/pose/main.py:294-296)
FLIP_PAIRS = {
    'coco': [(2, 1), (4, 3), (6, 5), (8, 7), (10, 9), (12, 11), (14, 13), (16, 15)],
    'mpii': [(0, 5), (1, 4), (2, 3), (10, 15), (11, 14), (12, 13)],
    'aic': [(0, 3), (1, 4), (2, 5), (6, 9), (7, 10), (8, 11)],
}
COCO_FLIP_PAIRS = FLIP_PAIRS['coco']


def get_flip_pairs(dataset):
    """Joint pairs swapped by the flip test for `dataset`; an unknown dataset raises (the reference only prints)."""
    if dataset not in FLIP_PAIRS:
        raise ValueError("flip test: no left/right joint table for dataset '{}'".format(dataset))
    return FLIP_PAIRS[dataset]


def _flip_back(hm, pairs):
    """Undo a horizontal flip of the INPUT on the heatmaps: mirror x and swap left/right channels."""
    hm = torch.flip(hm, dims=[3])
    idx = list(range(hm.shape[1]))
    for a, b in pairs:
        if a >= len(idx) or b >= len(idx):
            raise ValueError('flip pair ({}, {}) does not fit {} heatmap channels: wrong dataset table?'.format(a, b, len(idx)))
        idx[a], idx[b] = idx[b], idx[a]
    return hm[:, idx]


def synthetic_batches(n, batch, res, seed):
    """(inputs, meta) batches of N(0,1) crops with box centre / scale metadata (coco.py:110-113 geometry)."""
    H, W = res
    done = 0
    while done < n:
        b = min(batch, n - done)
        x = synth.pose_crops(seed + done, b, H, W)
        center = np.stack([np.array([W / 2.0 + 3 * i, H / 2.0 + 2 * i]) for i in range(b)])
        scale = np.array([max(H, W * H / W) * 1.25 for _ in range(b)], dtype=np.float64)
        yield x, {'center': center, 'scale': scale, 'index': np.arange(done, done + b)}
        done += b


def validate(model, batches, flip_test=False, adjust_coords=True, flip_pairs=COCO_FLIP_PAIRS, rank=0, world=1, device='cuda',
             preds_fn=None):
    """Inference loop of main.py:254-372 (eval mode, forward, flip test, final_preds). Returns dict of arrays.

    world > 1 (one process per GPU under torchrun, parallel.init_from_env): every rank walks the SAME batches, runs the net on its
    contiguous slice of each batch (parallel.shard_range) and the key points / scores of the slices are all-gathered
    (parallel.all_gather_rows: one collective per batch), so every rank returns the full arrays in dataset order — what the
    reference gets from nn.DataParallel's scatter / gather (tools/flownet/main.py:133-134,186).  `preds_fn(heatmaps, center,
    scale, adjust_coords)` defaults to the device path (evaluation.final_preds); `device` / `preds_fn` exist for the gloo test."""
    from flowtrack.pytorch_amd import parallel
    if hasattr(model, 'eval'):
        model.eval()
    preds_fn = preds_fn or evaluation.final_preds
    on_gpu = torch.device(device).type == 'cuda'
    all_preds, all_scores, all_idx = [], [], []
    n, t0 = 0, time.time()
    for inputs, meta in batches:
        B = inputs.shape[0]
        lo, hi = parallel.shard_range(B, rank, world)
        center, scale = np.asarray(meta['center'])[lo:hi], np.asarray(meta['scale'])[lo:hi]
        if hi > lo:
            x = inputs[lo:hi].to(device, non_blocking=True)
            output = model(x)
            if flip_test:
                flipped = _flip_back(model(torch.flip(x, dims=[3])), flip_pairs)
                output = (output + flipped) * 0.5
            preds, scores = preds_fn(output, center, scale, adjust_coords)
            rows = torch.from_numpy(np.concatenate((np.asarray(preds, np.float64), np.asarray(scores, np.float64)), axis=2))
        else:
            rows = None
        if world > 1:
            if B < world:           # more ranks than crops in this batch: the empty shards learn the joint count, then join the collective
                k = torch.tensor([0 if rows is None else rows.shape[1]], dtype=torch.int64, device=device if on_gpu else None)
                torch.distributed.all_reduce(k, op=torch.distributed.ReduceOp.MAX)
                if rows is None:
                    rows = torch.zeros((0, int(k.item()), 3), dtype=torch.float64)
            rows = parallel.all_gather_rows(rows.to(device) if on_gpu else rows, B).cpu()
        all_preds.append(rows[..., :2].numpy())
        all_scores.append(rows[..., 2:].numpy().astype(np.float32))
        all_idx.append(np.asarray(meta['index']))
        n += B
    if on_gpu:
        torch.cuda.synchronize()
    dt = time.time() - t0
    if rank == 0:
        print('validate: {} crops in {:.3f} s ({:.1f} crops/s incl. host post-processing{})'.format(
            n, dt, n / max(dt, 1e-9), ', sharded over {} ranks'.format(world) if world > 1 else ''))
    return {'preds': np.concatenate(all_preds), 'scores': np.concatenate(all_scores), 'index': np.concatenate(all_idx),
            'seconds': dt}


def main(**kwargs):
    opt.parse(kwargs)
    opt.work_dir = os.path.join(opt.checkpoint_path, opt.dataset, opt.exp_id)
    num_classes = num_joints[opt.dataset] + (1 if opt.with_bg else 0)
    print("==> creating model '{}', backbone = {}".format(opt.model, opt.backbone))
    if opt.model != 'deconv':
        raise ValueError("only model='deconv' (ResNet + 3-deconv head) is on the HIP path; got '{}'".format(opt.model))
    model = getattr(models, opt.model)(opt.backbone, num_classes=num_classes, pretrained=opt.pretrained)
    opt.model_name = '{}_{}'.format(opt.model, opt.backbone)
    if not opt.use_gpu:
        raise ValueError('the HIP path needs use_gpu=True (CPU reference results: oracle/)')
    # one process per GPU under torchrun (RANK / LOCAL_RANK / WORLD_SIZE): the batches are sharded in validate()
    from flowtrack.pytorch_amd import parallel
    rank, local_rank, world = parallel.init_from_env()
    if world > 1:
        torch.cuda.set_device(local_rank)
    model = model.cuda()
    if opt.resume:
        model_path = os.path.join(opt.work_dir, opt.resume)
        if os.path.exists(model_path):
            print("=> loading checkpoint '{}'".format(opt.resume))
            checkpoint = torch.load(model_path, map_location='cpu')
            model.load_state_dict(checkpoint['state_dict'])
            print("=> loaded checkpoint '{}' (epoch {})".format(opt.resume, checkpoint.get('epoch')))
        else:
            print("=> no checkpoint found at '{}'".format(opt.resume))
    print('    Total params: %.4fM' % (sum(p.numel() for p in model.parameters()) / 1000000.0))
    if opt.fp16:
        model.compute_dtype = torch.float16
    if 'valid' not in opt.run_type and 'test' not in opt.run_type:
        raise ValueError("run_type '{}': training is out of scope of the HIP path (use 'valid')".format(opt.run_type))
    batches = kwargs.get('batches') or synthetic_batches(opt.num_samples, opt.test_batch_size, opt.input_res, opt.seed)
    out = validate(model, batches, flip_test=opt.flip_test, adjust_coords=opt.adjust_coords,
                   flip_pairs=get_flip_pairs(opt.dataset) if opt.flip_test else (), rank=rank, world=world)
    out['model'] = model
    return out


if __name__ == '__main__':
    import ast
    kw = {}
    for a in sys.argv[1:]:
        k, v = a.lstrip('-').split('=', 1)
        try:
            kw[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            kw[k] = v
    main(**kw)
