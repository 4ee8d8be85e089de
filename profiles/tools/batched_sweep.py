"""Throughput of one launch against the number of chains in it, the speculative waves per chain and the point capacity
(= LDS per chain, hence chains resident per CU): 256-px tiles, 50 objects, 30 257 steps each."""
import json, os, sys, subprocess
out = {}
cases = [(256, 8, 128), (256, 1, 128), (4096, 1, 128), (16384, 1, 128)]
cases += [(t, s, c) for c in (128, 1024) for t in (512, 1024, 2048) for s in (8, 4, 2, 1)]
for tiles, spec, cap in cases:
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--no-convergence", "--no-cpu-baseline",
                        "--batched-tiles", str(tiles), "--batched-spec", str(spec), "--batched-capacity", str(cap)],
                       capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])["batched"]
    except Exception as e:                                    # keep sweeping; the failure is recorded
        out[f"{tiles} chains, spec_waves {spec}, capacity {cap}"] = {"error": (r.stderr or str(e))[-300:]}
        print(tiles, spec, cap, "FAILED", flush=True)
        continue
    out[f"{tiles} chains, spec_waves {spec}, capacity {cap}"] = {"proposals_per_s": d["proposals_per_s"], "kernel_ms": d["kernel_ms"]}
    print(tiles, spec, cap, round(d["proposals_per_s"] / 1e6, 1), "M/s", round(d["kernel_ms"], 1), "ms", flush=True)
    json.dump(out, open("gpurun_out/batched_sweep.json", "w"), indent=1)
