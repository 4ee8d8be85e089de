#!/bin/bash
# diagnostic build of the deep-round kernel with per-phase clock64 stamps (never shipped; MPP_DEEP_PROF in csrc/mpp_deep.hip)
set -e
cd "$(dirname "$0")/../../mpp_cnn_rs_object_detection_amd"
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -fvisibility=hidden -std=c++17 -Wno-unused-variable -Wno-unused-function"
/opt/rocm/bin/hipcc $F -mllvm -disable-machine-licm -mllvm -unroll-threshold=600 -mllvm -unroll-runtime -DMPP_DEEP_PROF -c csrc/mpp_deep.hip -o /tmp/mpp_deep_prof.o
OBJS=$(ls csrc/*.o | grep -v mpp_deep.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmppgpu_dprof.so $OBJS /tmp/mpp_deep_prof.o
echo built libmppgpu_dprof.so
