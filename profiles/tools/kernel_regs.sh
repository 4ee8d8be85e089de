#!/bin/bash
# registers / scratch of every instantiation of the chain kernel in a built object:
#   bash profiles/tools/kernel_regs.sh [path/to/mpp_sampler.o]
obj=${1:-mpp_cnn_rs_object_detection_amd/csrc/mpp_sampler.o}
tmp=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy -O binary --only-section=.hip_fatbin $obj $tmp/fat.bin
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.co --unbundle
$B/llvm-readelf --notes $tmp/dev.co | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in re.split(r"\n\s+- \.agpr_count", txt)[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or "mpp_chain_kernel" not in name.group(1): continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    targs = re.search(r"ILi(\d+)ELi(\d+)ELb(\d)ELi(\d+)ELb(\d)ELb(\d)E", name.group(1))
    print("WAVES,LPW,DIAG,OCC,SM,FAST =", ",".join(targs.groups()) if targs else name.group(1), " vgpr", g("vgpr_count"), "spill", g("vgpr_spill_count"), "sgpr", g("sgpr_count"), "scratch", g("private_segment_fixed_size"))
' | sort
rm -rf $tmp
