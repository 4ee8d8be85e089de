#!/bin/bash
# A/B of library variants on ONE box: bash profiles/tools/benchcmp.sh <variant> [<variant> ...]   ("main" = the shipped library;
# other names = libmppgpu_<name>.so built by profiles/tools/build_variant.sh; a trailing :nofast sets MPP_NO_FAST=1)
for v in "$@"; do
  name=${v%%:*}
  unset MPP_NO_FAST MPP_LIB_PATH
  if [ "$name" != main ]; then export MPP_LIB_PATH=$PWD/mpp_cnn_rs_object_detection_amd/libmppgpu_$name.so; fi
  case "$v" in *:nofast) export MPP_NO_FAST=1;; esac
  for rep in 1 2; do
    python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-convergence --scene 0 --mosaic 0 > gpurun_out/cmp_$name.json 2> gpurun_out/cmp_$name.err
    python - <<PY
import json; d=json.load(open("gpurun_out/cmp_$name.json")); print("$v rep $rep: single", round(d["value"]), "kernel_ms", round(d["roofline"]["kernel_ms"],2), "batched", round(d["batched"]["proposals_per_s"]/1e6,1), "M/s")
PY
  done
done
