#!/bin/bash
# diagnostic build with per-phase clock64 stamps (never shipped; see MPP_PROFILE in csrc/mpp_sampler.hip)
cd "$(dirname "$0")/../../mpp_cnn_rs_object_detection_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -disable-machine-licm -mllvm -unroll-threshold=600 -mllvm -unroll-runtime -fPIC -fvisibility=hidden -std=c++17 -DMPP_PROFILE -shared -o libmppgpu_prof.so csrc/*.hip
