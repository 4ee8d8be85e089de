"""Build libmppgpu.so (hipcc, gfx950 only) in-tree."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmppgpu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the CPU oracle forms no FMA, and parity of accept decisions is the contract
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-std=c++17", "-Wall", "-Wno-unused-function"]
# The chain kernel is one huge loop body; machine LICM hoists every float64 literal of the inlined exp/log/sincos
# polynomials into registers that then spill (296 VGPRs + 40 spills -> 258 VGPRs, 2 spills without it).
# -unroll-threshold=600 (the default is 300): the fixed-trip loops of the step -- the 32-class passes over a mark row, the
# four-edge / four-corner loops of the clipper, the Philox rounds -- unroll fully: +2.7 % on one tile, +3.2 % with 4 096
# chains on the same box (400: +2.7 % / -1.8 %; 1 200 and 2 500: as 600; -fno-unroll-loops: -10 %; -O2, -Os, the max-ilp and
# max-memory-clause schedulers: no gain or worse).  -unroll-runtime (loops whose trip count is only known at run time get an
# unrolled body with a remainder loop): +0.6 % / +0.7 % on top.  Same arithmetic, byte-identical chains.
_CHAIN_FLAGS = ["-mllvm", "-disable-machine-licm", "-mllvm", "-unroll-threshold=600", "-mllvm", "-unroll-runtime"]
EXTRA = {"mpp_sampler.hip": _CHAIN_FLAGS, "mpp_deep.hip": _CHAIN_FLAGS}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(os.path.dirname(HERE), "include", "mpp_hip.h"),
                                                                 os.path.abspath(__file__)]       # (the flags live here)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps()):
        return LIB
    objs, todo = [], []
    for src in sources():
        obj = os.path.splitext(src)[0] + ".o"
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d)
                                                   for d in [src] + deps()[len(sources()):]):
            todo.append([HIPCC] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj])
        objs.append(obj)
    if todo:                                            # (the two chain-kernel files take 1.5 minutes each: side by side)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(4, len(todo))) as pool:
            list(pool.map(run, todo))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
