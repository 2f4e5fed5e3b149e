# dev A/B (one gpurun call = one box): the pose stem reading the NCHW fp32 input itself (FT_FUSE_STEM_PACK=0 = ft_pack_nchw_to_nhwc in front)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_pose_gpu.py -x -q -k "stem" 2>&1 | tail -4
for i in 1 2; do for f in 0 1; do
  echo "== FT_FUSE_STEM_PACK=$f"; FT_FUSE_STEM_PACK=$f timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 400 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
FT_FUSE_STEM_PACK=0 timeout 300 python tools/dev/net_bench.py resnet50 64 256 192 fp16 2>&1 | grep -E "^ +[0-2] |graph replay"
FT_FUSE_STEM_PACK=1 timeout 300 python tools/dev/net_bench.py resnet50 64 256 192 fp16 2>&1 | grep -E "^ +[0-2] |graph replay"
