#!/bin/bash
# build a variant of the library for A/B measurements: bash profiles/tools/build_variant.sh <name> [extra hipcc flags...]
# -> mpp_cnn_rs_object_detection_amd/libmppgpu_<name>.so (select it with MPP_LIB_PATH); never shipped
set -e
name=$1; shift
cd "$(dirname "$0")/../../mpp_cnn_rs_object_detection_amd"
tmp=$(mktemp -d)
for f in csrc/*.hip; do
  extra=""
  if [ "$(basename $f)" = mpp_sampler.hip ]; then extra="-mllvm -disable-machine-licm -mllvm -unroll-threshold=600 -mllvm -unroll-runtime"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -fvisibility=hidden -std=c++17 $extra "$@" -c $f -o $tmp/$(basename $f .hip).o &
done
wait
for f in csrc/*.hip; do test -s $tmp/$(basename $f .hip).o || { echo "compile of $f failed"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmppgpu_$name.so $tmp/*.o
rm -rf $tmp
ls -la libmppgpu_$name.so
