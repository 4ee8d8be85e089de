"""Steady-state kernel breakdown of the score-map forward (torch.profiler, after warm-up)."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from mpp_cnn_rs_object_detection_amd import unet
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
img = torch.rand((size, size, 3))
for dtype in (torch.float32, torch.bfloat16):
    runner = unet.ScoreMapNets(unet.PosNet(), unet.ShapeNet(), device=0, dtype=dtype)
    for _ in range(4):
        runner.infer(img)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            runner.infer(img)
        torch.cuda.synchronize()
    print("=====", dtype, size)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=70))
