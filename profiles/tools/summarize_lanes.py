#!/usr/bin/env python3
"""Condense gpurun_out/prof_lanes_<tag>/ (profiles/tools/pmc_lanes.sh) into a markdown table: per launch of the chain
kernel, VALU instructions, busy fraction and the mean number of active lanes per VALU instruction
(SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, rocprofiler-sdk's own `VALUThreadUtilization`-style ratio)."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", f"prof_lanes_{tag}")
GHZ = 2.4
print(f"# lane occupancy of `mpp_chain_kernel` ({tag})\n")
print("Source: `bash profiles/tools/pmc_lanes.sh %s` on one MI355X; one --pmc pass per launch kind, --kernel-trace only.\n" % tag)
summary = {}
for kind in ("one", "many"):
    rows = {}
    for p in sorted(glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv"))):
        kt = p.replace("_counter_collection.csv", "_kernel_trace.csv")
        dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
               for r in csv.DictReader(open(kt)) if ("mpp_chain" in r["Kernel_Name"] or "mpp_deep" in r["Kernel_Name"])}
        agg = collections.defaultdict(dict)
        for r in csv.DictReader(open(p)):
            if ("mpp_chain" in r["Kernel_Name"] or "mpp_deep" in r["Kernel_Name"]):
                agg[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        if not agg:
            continue
        # the longest PRODUCTION launch of the run (template argument DIAG = false: "<W, 0, false, ...>"), not the traced chain
        prod = [k for k in agg if re.search(r"<\d+, (\d+, )?false,", dur.get(k, (0, ""))[1])] or list(agg)
        d = max(prod, key=lambda k: dur.get(k, (0, ""))[0])
        rows = dict(agg[d], ns=dur[d][0], name=dur[d][1])
        # a chain that starts hot is two launches: the one-wave-per-step kernel right before the deep-round launch belongs
        # to the same chain (its counters and its time are added)
        before = sorted((k for k in prod if int(k) < int(d)), key=int)
        if "mpp_deep" in dur[d][1] and before and "mpp_chain_kernel" in dur[before[-1]][1]:
            h = before[-1]
            for k, v in agg[h].items():
                rows[k] = rows.get(k, 0.0) + v
            rows["ns"] += dur[h][0]
            rows["name"] = dur[d][1] + " + the hot start's " + dur[h][1][:40]
    if not rows:
        print(f"## {kind}: no chain-kernel dispatch found\n")
        continue
    bj = os.path.join(src, f"bench_{kind}.json")
    bench = {}
    if os.path.exists(bj) and os.path.getsize(bj):
        try:
            bench = json.loads(open(bj).read().strip().splitlines()[-1])
        except Exception:
            bench = {}
    if kind == "one":
        proposals = bench.get("config", {}).get("iters_per_step", 100001)
        n_simd = 4
    else:
        b = bench.get("batched", {})
        proposals = b.get("tiles", 4096) * b.get("iters", 30257)
        n_simd = 1024
    ns = rows["ns"]
    cyc = ns * GHZ
    act, thr, inst = rows.get("SQ_ACTIVE_INST_VALU", 0.0), rows.get("SQ_THREAD_CYCLES_VALU", 0.0), rows.get("SQ_INSTS_VALU", 0.0)
    icyc = rows.get("SQ_INST_CYCLES_VALU", 0.0)
    print(f"## {'one 512x512 tile (1 workgroup, 8 waves)' if kind == 'one' else 'many chains in one launch'}\n")
    print(f"kernel `{rows['name'][:80]}`, {ns / 1e6:.2f} ms, {proposals} proposals\n")
    print("| quantity | value |\n|---|---|")
    for k in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU",
              "SQ_INST_CYCLES_VALU", "SQ_THREAD_CYCLES_VALU"):
        if k in rows:
            print(f"| {k} | {rows[k]:.5g} |")
    if inst:
        print(f"| VALU instructions per proposal | {inst / proposals:.0f} |")
        print(f"| SALU instructions per proposal | {rows.get('SQ_INSTS_SALU', 0) / proposals:.0f} |")
    if act:
        print(f"| VALU busy per SIMD over the SIMDs in use ({n_simd}): ACTIVE_INST_VALU x4 / SIMDs / cycles @2.4 GHz | {act * 4 / n_simd / cyc:.3f} |")
        print(f"| issue-slot fraction: VALU instructions / (SIMDs x cycles) | {inst / (n_simd * cyc):.4f} |")
    if act and thr:
        print(f"| **active lanes per VALU cycle: THREAD_CYCLES_VALU / ACTIVE_INST_VALU** | **{thr / act:.2f} of 64** |")
    if icyc and thr:
        print(f"| active lanes per VALU cycle: THREAD_CYCLES_VALU / INST_CYCLES_VALU | {thr / icyc:.2f} of 64 |")
    if act and inst:
        print(f"| quad-cycles per VALU instruction: ACTIVE_INST_VALU / INSTS_VALU | {act / inst:.2f} |")
    print()
    summary[kind] = {"kernel": rows["name"].split("(")[0], "kernel_ms_under_profiler": ns / 1e6, "proposals": proposals,
                     "valu_instructions_per_proposal": inst / proposals, "salu_instructions_per_proposal": rows.get("SQ_INSTS_SALU", 0) / proposals,
                     "valu_busy_frac_of_simds_in_use": act * 4 / n_simd / cyc if act else None, "simds_in_use": n_simd,
                     "exec_lanes_per_valu_cycle": thr / act if act and thr else None}
with open(os.path.join(src, "lanes.json"), "w") as f:
    json.dump(dict(summary, source=f"profiles/tools/pmc_lanes.sh {tag} (rocprofv3 --pmc, one pass per launch kind)"), f, indent=1)
