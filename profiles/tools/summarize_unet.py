#!/usr/bin/env python3
"""Summary of the rocprofv3 runs of profiles/tools/pmc_unet4096.sh: python3 profiles/tools/summarize_unet.py TAG
(reads gpurun_out/prof_unet4096_TAG, writes profiles/TAG_unet_pmc.md)."""
import csv, glob, collections, json, os, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
OUT = os.path.join(os.getcwd(), "gpurun_out", f"prof_unet4096_{TAG}")
out = [f"# rocprofv3: the 4096 x 4096 float32 score-map forward ({TAG})", "",
       f"Source: `bash profiles/tools/pmc_unet4096.sh {TAG}` on one MI355X: `profiles/tools/prof_unet_4096.py` (PosNet + ShapeNet + epilogues, "
       "channels-last, random-init weights, 5 forwards per run), --kernel-trace --stats, then separate --pmc passes.", ""]
try:
    out += ["The forward as the product runs it (two streams, no profiler): `" + open(f"{OUT}/bench_two_streams.json").read().strip().splitlines()[-1] + "`", ""]
except Exception:
    pass
try:
    out += ["Bench line under the profiler (ONE stream, `MPP_UNET_TWO_STREAMS=0`, as are all tables below): `" + open(f"{OUT}/bench.json").read().strip().splitlines()[-1] + "`", ""]
except Exception as e:
    out += [f"(no bench line: {e})", ""]
# only the LAST 3 forwards count (the first passes pay MIOpen's algorithm search: naive reference convolutions and
# benchmark runs of every candidate solver); a forward ends with the fused shapenet heads
def last_forwards(path, n_last=3):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    ends = [int(r["End_Timestamp"]) for r in rows if "k_shapenet_heads" in r["Kernel_Name"]]
    if not ends:                       # (the unfused heads: a forward ends with the third softmax epilogue)
        ends = [int(r["End_Timestamp"]) for r in rows if "k_shapenet_epilogue" in r["Kernel_Name"]][2::3]
    t0 = ends[-n_last - 1] if len(ends) > n_last else 0
    return [r for r in rows if int(r["Start_Timestamp"]) >= t0], n_last if len(ends) > n_last else max(1, len(ends))
dur = collections.defaultdict(float); calls = collections.Counter(); nf = 3; keep_ids = {}
for p in glob.glob(f"{OUT}/trace/*/*_kernel_trace.csv"):
    rows, nf = last_forwards(p)
    for r in rows:
        k = r["Kernel_Name"][:90]; dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; calls[k] += 1
tot = sum(dur.values())
out += [f"## kernels of the last {nf} forwards (ms)", "", "| kernel | calls | total ms | % |", "|---|---|---|---|"]
for k, v in sorted(dur.items(), key=lambda kv: -kv[1])[:25]:
    out.append(f"| `{k}` | {calls[k]} | {v:.2f} | {100 * v / tot:.1f} |")
out += ["", f"total kernel time {tot:.1f} ms over {nf} forwards = {tot / nf:.1f} ms per forward", ""]
# the launches of the last forward in order (what precedes what: zero fills, bias adds, layout copies)
for p in glob.glob(f"{OUT}/trace/*/*_kernel_trace.csv"):
    rows, _ = last_forwards(p, 1)
    out += ["## the last forward, launch by launch", "", "| # | kernel | grid | ms |", "|---|---|---|---|"]
    for i, r in enumerate(rows):
        out.append(f"| {i} | `{r['Kernel_Name'][:70]}` | {r.get('Grid_Size', '')} | {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.3f} |")
    out.append("")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob(f"{OUT}/pmc*/*/*_counter_collection.csv"):
    kt = p.replace("_counter_collection.csv", "_kernel_trace.csv")
    ids = {r["Dispatch_Id"] for r in last_forwards(kt)[0]}
    for r in csv.DictReader(open(p)):
        if r["Dispatch_Id"] in ids:
            agg[r["Kernel_Name"][:90]][r["Counter_Name"]] += float(r["Counter_Value"])
out += ["## counters, summed over the same forwards, for the kernels above", "",
        "| kernel | SQ_VALU_MFMA_BUSY_CYCLES | SQ_BUSY_CYCLES | MFMA busy cycles per SQ busy cycle / 4 | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM GB/s (2 x FETCH + WRITE over the kernel's time) |", "|---|---|---|---|---|---|---|"]
for k, v in sorted(dur.items(), key=lambda kv: -kv[1])[:25]:
    a = agg.get(k, {})
    mf, sq, fe, wr = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), a.get("SQ_BUSY_CYCLES", 0), a.get("FETCH_SIZE", 0), a.get("WRITE_SIZE", 0)
    out.append(f"| `{k[:60]}` | {mf:.4g} | {sq:.4g} | {mf / (4 * sq) if sq else 0:.3f} | {fe:.4g} | {wr:.4g} | {(2 * fe + wr) * 1024 / (v * 1e-3) / 1e9 if v else 0:.0f} |")
out += ["", "FETCH_SIZE doubled for wide coalesced streams (gfx950 correction of the microarchitecture guide), WRITE_SIZE as is; KiB units.", ""]
open(f"profiles/{TAG}_unet_pmc.md", "w").write("\n".join(out))
print("\n".join(out))
