"""FlowNet demo entry point on the HIP path — same CLI as the reference's tools/flownet/demo.py:22-99.

    python tools/flownet/demo.py --model FlowNet2S --resume CKPT.pth.tar -i img0.ppm -p img1.ppm -s results [--fp16] [-ng 1]

Builds `models.<model>(args, **model_kwargs)` through the same argument reflection (`--model_<kw>` flags),
loads `checkpoint['state_dict']`, packs the pair as [1,3,2,H,W] RGB 0..255, runs the HIP forward and writes
`<save>/output.flo` + `<save>/flow.png`.  Differences: images are read with PIL (scipy.misc.imread is gone);
frames whose size is not a multiple of 64 are edge-replicated bottom/right (pack_pair) and the flow cropped back;
`--random_weights` allows a run without a checkpoint (the reference quits, demo.py:57-59);
`--number_gpus > 1` is accepted but the single pair runs on one GPU.  Batches of pairs shard one process per GPU:

    torchrun --nproc-per-node 8 tools/flownet/demo.py --model FlowNet2S --resume CKPT --pair_list pairs.txt -s results

(`--pair_list`: one `img0 img1` per line, all of one size; every rank runs its contiguous slice of the list (run_pairs), writes
`<save>/<k>.flo` for the pairs it owns and the per-pair flow statistics are all-gathered so that rank 0 prints the whole table —
the role of the reference's nn.DataParallel scatter / gather in tools/flownet/main.py:133-134,186,197.)
"""
from __future__ import absolute_import, division, print_function

import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from flowtrack.pytorch_amd.flownet import models, tools   # noqa: E402

MODEL_CHOICES = ['FlowNet2', 'FlowNet2S', 'FlowNet2C', 'FlowNet2CS', 'FlowNet2CSS', 'FlowNet2SD']


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--input1', '-i', default='samples/img0.ppm', type=str, help='first input image')
    parser.add_argument('--input2', '-p', default='samples/img1.ppm', type=str, help='second input image')
    parser.add_argument('--save', '-s', default='results', type=str, help='directory for saving')
    parser.add_argument('--resume', default='', type=str, metavar='PATH', help='path to latest checkpoint (default: none)')
    parser.add_argument('--number_gpus', '-ng', type=int, default=1, help='number of GPUs to use')
    parser.add_argument('--fp16', action='store_true', help='Run model in pseudo-fp16 mode (fp16 storage fp32 math).')
    parser.add_argument('--rgb_max', type=float, default=255.)
    parser.add_argument('--random_weights', action='store_true', help='run without a checkpoint (results are meaningless)')
    parser.add_argument('--pair_list', default='', type=str, help='text file, one "img0 img1" per line: batch mode, sharded over the '
                        'ranks of a torchrun launch')
    parser.add_argument('--batch_size', '-b', type=int, default=4, help='pairs per forward in batch mode')
    tools.add_arguments_for_module(parser, models, argument_for_class='model', default='FlowNet2S', choices=MODEL_CHOICES)
    return parser


def load_image(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))


def pack_pair(im1, im2):
    """im1, im2: HxWx3 RGB -> fp32 tensor [1,3,2,Hp,Wp], Hp / Wp the next multiples of 64, filled by REPLICATING the last row /
    column — the one helper the tracking glue uses too (flowtrack.pytorch_amd.tracking.net_utils.pad_pairs_to_64): zero padding
    would put a hard black edge into the frame and pull the network's own rgb_mean (lib/flownet/model/models.py:255) towards
    black for 1080p-type inputs.  The reference's demo feeds frames as they are and only works for sizes the net divides."""
    from flowtrack.pytorch_amd.tracking.net_utils import pad_pairs_to_64
    pair = np.stack((np.asarray(im1, dtype=np.float32), np.asarray(im2, dtype=np.float32)))    # [2,H,W,3]
    return pad_pairs_to_64(torch.from_numpy(pair).permute(3, 0, 1, 2).unsqueeze(0).contiguous())


def run_pair(model, im1, im2):
    """im1, im2: HxWx3 uint8/float RGB -> flow HxWx2 fp32 (pixels)."""
    H, W = im1.shape[:2]
    with torch.no_grad():
        flow = model(pack_pair(im1, im2).cuda()).cpu()
    return flow[0, :, :H, :W].numpy().transpose(1, 2, 0)


def run_pairs(model, pairs, rank=0, world=1, batch=4, device='cuda', on_flow=None):
    """pairs: list of (im1, im2) HxWx3 arrays of one size.  Rank `rank` of `world` runs its contiguous slice (parallel.shard_range)
    in batches of `batch`, hands every flow field to `on_flow(global_index, flow HxWx2)` and returns the [len(pairs), 4] table
    (index, mean u, mean v, max |flow|) of ALL pairs, all-gathered so that every rank holds it (one collective)."""
    from flowtrack.pytorch_amd import parallel
    n = len(pairs)
    lo, hi = parallel.shard_range(n, rank, world)
    rows = []
    for b0 in range(lo, hi, batch):
        chunk = pairs[b0:min(b0 + batch, hi)]
        H, W = chunk[0][0].shape[:2]
        x = torch.cat([pack_pair(a, b) for a, b in chunk], 0).to(device)
        with torch.no_grad():
            flow = model(x).float().cpu()
        for k in range(len(chunk)):
            f = flow[k, :, :H, :W].numpy().transpose(1, 2, 0)
            if on_flow is not None:
                on_flow(b0 + k, f)
            rows.append([float(b0 + k), float(f[..., 0].mean()), float(f[..., 1].mean()), float(np.sqrt((f ** 2).sum(-1)).max())])
    local = torch.tensor(rows, dtype=torch.float64).reshape(-1, 4)
    if world > 1:
        on_gpu = torch.device(device).type == 'cuda'
        local = parallel.all_gather_rows(local.to(device) if on_gpu else local, n).cpu()
    return local.numpy()


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    args.model_class = tools.module_to_dict(models)[args.model]
    kwargs = tools.kwargs_from_args(args, 'model')
    model = args.model_class(args, **kwargs)
    print('Building {} model: {} parameters'.format(args.model, sum(p.numel() for p in model.parameters())))
    if args.resume and os.path.isfile(args.resume):
        print("Loading checkpoint '{}'".format(args.resume))
        checkpoint = torch.load(args.resume, map_location='cpu')
        model.load_state_dict(checkpoint['state_dict'])
        print("Loaded checkpoint '{}' (at epoch {}, best_EPE {})".format(args.resume, checkpoint.get('epoch'), checkpoint.get('best_EPE')))
    elif args.random_weights:
        print('No checkpoint: running with randomly initialised weights (--random_weights)')
    else:
        print("No checkpoint found at '{}'".format(args.resume))
        return 1
    os.makedirs(args.save, exist_ok=True)
    if args.number_gpus < 1:
        raise SystemExit('the HIP path needs a GPU (--number_gpus >= 1)')
    if args.number_gpus > 1:
        # the reference wraps the model in nn.DataParallel (tools/flownet/demo.py:75), which scatters dim 0: a one-pair batch
        # runs on GPU 0 there as well.  Here multi-GPU is one process per GPU (torchrun + bench.py / parallel.py).
        print('[demo] --number_gpus %d: one frame pair cannot be sharded, it runs on GPU 0 (as under the reference\'s '
              'DataParallel); batches shard one process per GPU via torchrun, see bench.py' % args.number_gpus, file=sys.stderr)
    from flowtrack.pytorch_amd import parallel
    rank, local_rank, world = parallel.init_from_env()
    if world > 1:
        torch.cuda.set_device(local_rank)
    model = model.cuda()
    if args.fp16:
        model = model.half()
    model.eval()
    if args.pair_list:
        with open(args.pair_list) as f:
            names = [ln.split() for ln in f if ln.strip()]
        pairs = [(load_image(a), load_image(b)) for a, b in (nm[:2] for nm in names)]
        table = run_pairs(model, pairs, rank, world, args.batch_size,
                          on_flow=lambda k, fl: tools.write_flow(fl, os.path.join(args.save, '%06d.flo' % k)))
        if rank == 0:
            for k, mu, mv, mx in table:
                print('%06d  mean flow (%.3f, %.3f) px  max |flow| %.3f px' % (int(k), mu, mv, mx))
            print('wrote %d flow fields to %s (%d rank%s)' % (len(pairs), args.save, world, 's' if world > 1 else ''))
        return 0
    if world > 1 and rank != 0:
        return 0            # the single-pair demo runs on rank 0 only
    flow = run_pair(model, load_image(args.input1), load_image(args.input2))
    tools.write_flow(flow, os.path.join(args.save, 'output.flo'))
    from PIL import Image
    Image.fromarray(tools.flow_to_image(flow)).save(os.path.join(args.save, 'flow.png'))
    print('wrote {0}/output.flo and {0}/flow.png ({1}x{2})'.format(args.save, flow.shape[1], flow.shape[0]))
    return 0


if __name__ == '__main__':
    sys.exit(main())
