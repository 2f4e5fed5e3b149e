"""Entry points: tools/pose/main.py main(**kwargs) and tools/flownet/demo.py CLI, end to end on the GPU."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.flownet import models as flow_models, tools as flow_tools
from flowtrack.pytorch_amd.pose import models as pose_models
from oracle import flow_ref, keypoints_ref, pose_ref

pytestmark = pytest.mark.gpu


def test_pose_main_resume_and_flip_test(hip_lib, tmp_path):
    from tools.pose.main import main, synthetic_batches
    m = pose_models.deconv("resnet50", 17, False)
    sd = synth.fill_pose_state_dict(m.state_dict(), 31)
    work = tmp_path / "coco" / "exp"
    work.mkdir(parents=True)
    torch.save({"epoch": 7, "model": "deconv_resnet50", "state_dict": sd, "best_loss": 0.5, "optimizer": {}}, work / "deconv_resnet50_best.pth")
    out = main(model="deconv", backbone="resnet50", dataset="coco", input_res=(256, 192), checkpoint_path=str(tmp_path),
               exp_id="exp", resume="deconv_resnet50_best.pth", num_samples=6, test_batch_size=4, flip_test=False, adjust_coords=True, seed=3)
    assert out["preds"].shape == (6, 17, 2) and out["scores"].shape == (6, 17, 1)
    # same crops through the CPU oracle + reference-semantics final_preds
    got_i = 0
    for x, meta in synthetic_batches(6, 4, (256, 192), 3):
        hm = pose_ref.pose_forward(sd, x).numpy()
        want, wscores, _, _ = keypoints_ref.final_preds_ref(hm, meta["center"], meta["scale"], adjust_coords=True)
        b = x.shape[0]
        assert np.allclose(out["preds"][got_i:got_i + b], want, atol=1e-2)
        assert np.allclose(out["scores"][got_i:got_i + b], wscores, atol=1e-3)
        got_i += b
    # flip test = average with the mirrored pass, left/right channels swapped back (main.py:289-299)
    out2 = main(model="deconv", backbone="resnet50", dataset="coco", input_res=(256, 192), checkpoint_path=str(tmp_path),
                exp_id="exp", resume="deconv_resnet50_best.pth", num_samples=2, test_batch_size=2, flip_test=True, seed=3)
    x = synth.pose_crops(3, 2)
    from tools.pose.main import COCO_FLIP_PAIRS
    hm = pose_ref.pose_forward(sd, x)
    hf = torch.flip(pose_ref.pose_forward(sd, torch.flip(x, dims=[3])), dims=[3])
    idx = list(range(17))
    for a, b in COCO_FLIP_PAIRS:
        idx[a], idx[b] = idx[b], idx[a]
    avg = ((hm + hf[:, idx]) * 0.5).numpy()
    _, wscores, _ = keypoints_ref.max_preds_ref(avg)
    assert np.allclose(out2["scores"], wscores, atol=1e-3)


def test_flownet_demo_cli(hip_lib, tmp_path):
    from PIL import Image
    from tools.flownet import demo
    G = np.load(os.path.join(GOLDEN, "flow_golden.npz"))
    pair = G["sample_pair_u8"]                      # crop of the reference's samples/img0.ppm, img1.ppm
    Image.fromarray(pair[0]).save(tmp_path / "img0.ppm")
    Image.fromarray(pair[1]).save(tmp_path / "img1.ppm")
    m = flow_models.FlowNet2S(types.SimpleNamespace(rgb_max=255.0, fp16=False))
    sd = synth.fill_flow_state_dict(m.state_dict(), int(G["seed"]))
    ckpt = tmp_path / "FlowNet2-S_checkpoint.pth.tar"
    torch.save({"arch": "FlowNet2S", "epoch": 1, "state_dict": sd, "best_EPE": 1.0}, ckpt)
    rc = demo.main(["--model", "FlowNet2S", "--resume", str(ckpt), "-i", str(tmp_path / "img0.ppm"), "-p", str(tmp_path / "img1.ppm"),
                    "-s", str(tmp_path / "results")])
    assert rc == 0
    flow = flow_tools.read_flow(str(tmp_path / "results" / "output.flo"))
    assert flow.shape == (256, 256, 2)
    assert np.abs(flow.transpose(2, 0, 1)[None] - G["sample_flow"]).max() <= 1e-3      # == the imported reference's output
    assert os.path.getsize(tmp_path / "results" / "flow.png") > 0
    assert demo.main(["--model", "FlowNet2S", "--resume", "/nonexistent", "-s", str(tmp_path / "r2")]) == 1   # reference quits here too
    # a frame that is not a multiple of 64: padded, flow cropped back
    f = demo.run_pair(m.cuda().eval(), pair[0][:200, :180], pair[1][:200, :180])
    assert f.shape == (200, 180, 2) and np.isfinite(f).all()
