#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (rocprofv3 csv output of profiles/run_profile.sh)
into profiles/<tag>_summary.md: per-kernel stats, PMC counters of the chain kernel, and the
bench JSON line measured under the profiler."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", f"prof_{tag}")
out = [f"# rocprofv3 summary `{tag}`", "", f"Source: `profiles/run_profile.sh {tag} ...` on one MI355X (gfx950).", ""]
for p in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    out += ["## kernel-trace --stats", "", "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(p)):
        out.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                   f"{float(r['AverageNs'])/1e6:.4f} | {float(r['Percentage']):.3f} |")
    out.append("")
out += ["## PMC (mean per CHAIN of the bench -- a chain that starts hot is two launches, mpp_chain_kernel then mpp_deep_kernel: "
        "their counters are added; separate passes)", "", "| counter | mean per chain |", "|---|---|"]
vals = {}
for p in sorted(glob.glob(os.path.join(src, "pmc*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(float)
    chains = collections.defaultdict(set)
    for r in csv.DictReader(open(p)):
        if ("mpp_chain" in r["Kernel_Name"] or "mpp_deep" in r["Kernel_Name"]):
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            if "mpp_deep" in r["Kernel_Name"]:
                chains[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k, v in agg.items():
        vals[k] = v / max(1, len(chains[k]))
        out.append(f"| {k} | {vals[k]:.6g} |")
out.append("")
if "FETCH_SIZE" in vals:
    # guide: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream
    out += [f"HBM traffic per chain: FETCH_SIZE {vals['FETCH_SIZE']:.0f} KiB (x2 gfx950 correction for wide streams = "
            f"{2*vals['FETCH_SIZE']/1024:.1f} MiB upper figure), WRITE_SIZE {vals.get('WRITE_SIZE', 0):.0f} KiB.", ""]
bj = os.path.join(src, "bench_trace.json")
if os.path.exists(bj) and os.path.getsize(bj):
    out += ["## bench line under the profiler", "", "```json", open(bj).read().strip(), "```", ""]
open(os.path.join("profiles", f"{tag}_summary.md"), "w").write("\n".join(out))
print("\n".join(out))
