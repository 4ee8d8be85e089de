import os, sys
import torch
sys.path.insert(0, os.getcwd())
from mpp_cnn_rs_object_detection_amd import unet
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(0)
size = 2048
img = torch.rand((size, size, 3))
runner = unet.ScoreMapNets(unet.PosNet(), unet.ShapeNet(), device=0, dtype=torch.float32)
for _ in range(4):
    runner.infer(img)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    runner.infer(img)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=False).table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=50))
x = torch.rand(1, 32, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
y = torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect")
print("pad reflect keeps channels_last:", y.is_contiguous(memory_format=torch.channels_last), y.stride())
