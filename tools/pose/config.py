"""Options of the pose entry point: same attribute names / kwargs protocol as the reference's
tools/pose/config.py:5-140 (class attributes overridden by `opt.parse(kwargs)`, unknown keys warn).
Defaults are the COCO / deconv / ResNet-50 / 256x192 set the north-star configs use (the reference
checks in mpii / fpn / densenet121 defaults and keeps this block commented out, config.py:8-62)."""
import warnings


class DefaultConfig(object):
    # general
    dataset = 'coco'
    data_path = './data/coco'
    num_workers = 4
    use_gpu = True
    # data
    input_res = (256, 192)        # (H, W)
    sigma = 2
    # model
    model = 'deconv'
    backbone = 'resnet50'
    stride = 4
    with_logits = False
    with_mask = False
    with_bg = False
    target_type = 'gaussian'
    pretrained = False
    tensorboard = False
    # run
    debug = False
    batch_size = 32
    test_batch_size = 32
    start_epoch = 0
    max_epoch = 140
    # checkpoint
    checkpoint_path = './checkpoints'
    exp_id = 'pretrained_01'
    resume = None                 # file name inside <checkpoint_path>/<dataset>/<exp_id>/
    run_type = 'valid'            # only the inference subset runs on the HIP path
    # evaluation
    adjust_coords = True
    flip_test = False
    oks_threshold = 0.9
    kpt_threshold = 0.2
    # HIP path
    fp16 = False                  # fp16 storage / fp32 accumulate (new capability; the reference pose tool is fp32 only)
    num_samples = 64              # synthetic crops to run when no dataset is given
    seed = 0


def parse(self, kwargs):
    for k, v in kwargs.items():
        if not hasattr(self, k):
            warnings.warn("Warning: opt has not attribut %s" % k)
        setattr(self, k, v)
    print('user config:')
    for k in sorted(set(dir(self.__class__)) - set(dir(object))):
        if not k.startswith('__') and not callable(getattr(self, k)):
            print(k, getattr(self, k))


DefaultConfig.parse = parse
opt = DefaultConfig()
