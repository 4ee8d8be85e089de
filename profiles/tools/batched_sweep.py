"""Throughput of one launch against the number of chains in it (256-px tiles, 50 objects, 30 257 steps each)."""
import json, os, sys, subprocess
out = {}
for tiles, spec in ((256, 8), (256, 1), (1024, 1), (4096, 1), (16384, 1)):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--no-convergence", "--no-cpu-baseline",
                        "--batched-tiles", str(tiles), "--batched-spec", str(spec), "--batched-capacity", "128"],
                       capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])["batched"]
    out[f"{tiles} chains, spec_waves {spec}"] = {"proposals_per_s": d["proposals_per_s"], "kernel_ms": d["kernel_ms"]}
    print(tiles, spec, d["proposals_per_s"], d["kernel_ms"], flush=True)
json.dump(out, open("gpurun_out/batched_sweep.json", "w"), indent=1)
