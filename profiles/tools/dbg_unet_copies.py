import os, sys, torch
sys.path.insert(0, os.getcwd())
from mpp_cnn_rs_object_detection_amd import unet
S = 1024
img = torch.rand((S, S, 3))
r = unet.ScoreMapNets(unet.PosNet(), unet.ShapeNet(), device=0)
orig = unet.ScoreMapNets._cl
def cl(x):
    if not x.is_contiguous(memory_format=torch.channels_last):
        print("COPY", tuple(x.shape), x.stride())
    return orig(x)
unet.ScoreMapNets._cl = staticmethod(cl)
r.infer(img)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    r.infer(img); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
