#!/usr/bin/env python3
"""Score-map forward on one MI355X: the two U-Nets (PyTorch-ROCm / MIOpen) and the two fused HIP epilogues.
Random-init weights (no trained model.pt in the container).  FLOP figure: 568 896 FLOP/pixel."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from mpp_cnn_rs_object_detection_amd import unet  # noqa: E402

FLOP_PER_PX = 568896
res = {}
torch.manual_seed(0)
for size in (512, 2048):
    img = torch.rand((size, size, 3))
    for dtype, layout in ((torch.float32, "nhwc"), (torch.bfloat16, "nhwc"), (torch.float32, "nchw"), (torch.bfloat16, "nchw")):
        runner = unet.ScoreMapNets(unet.PosNet(), unet.ShapeNet(), device=0, dtype=dtype, layout=layout)
        for _ in range(3):
            det, marks = runner.infer(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            det, marks = runner.infer(img)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[f"forward+epilogue {size}x{size} {str(dtype).split('.')[-1]} {layout}"] = {
            "ms": dt * 1e3, "Mpx_per_s": size * size / dt / 1e6, "TFLOP_per_s": FLOP_PER_PX * size * size / dt / 1e12}
    # epilogues alone (HBM-bound): bytes = read + write
    H = W = size
    pos_out = torch.randn((3, H, W), device="cuda")
    logits = torch.randn((32, H, W), device="cuda")
    det = torch.empty((H, W), device="cuda")
    m = torch.empty((H, W, 32), device="cuda")
    runner.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for name, fn, nbytes in (("posnet_epilogue", lambda: runner.ctx.posnet_epilogue(pos_out, H, W, -10.8, -2.1, det), 16 * H * W),
                             ("shapenet_epilogue", lambda: runner.ctx.shapenet_epilogue(logits, H, W, m), 256 * H * W)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res[f"{name} {size}x{size}"] = {"ms": ms, "GB_per_s": nbytes / ms / 1e6, "frac_of_8TBs": nbytes / ms / 1e6 / 8000}
    # the channels-last glue alone (HBM-bound): bytes = read + write of a 32-channel level
    for dtype in (torch.float32, torch.bfloat16):
        eb = 4 if dtype == torch.float32 else 2
        x = torch.randn((1, H, W, 32), device="cuda").to(dtype).permute(0, 3, 1, 2)
        sc, sh = torch.rand(32, device="cuda"), torch.rand(32, device="cuda")
        out = torch.empty((1, H + 2, W + 2, 32), device="cuda", dtype=dtype).permute(0, 3, 1, 2)
        fn = lambda: runner.ctx.nhwc_glue(x, pad=1, scale=sc, shift=sh, out=out)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        nbytes = eb * 32 * (H * W + (H + 2) * (W + 2))
        res[f"nhwc_glue pad+affine+relu 32ch {size}x{size} {str(dtype).split('.')[-1]}"] = {
            "ms": ms, "GB_per_s": nbytes / ms / 1e6, "frac_of_8TBs": nbytes / ms / 1e6 / 8000}
print(json.dumps(res, indent=1))
