#!/usr/bin/env python3
"""Turns the rocprofv3 output of scripts/profile_valu.sh (gpurun_out/<tag>/) into the tracked evidence under profiles/:

  profiles/<tag>_kernel_stats_bench.csv          rocprofv3 --kernel-trace --stats of the default bench command (under the
                                                 profiler the payload's device-to-host copies run as blit kernels and slow
                                                 the analysis kernel beside them: see scripts/profile_valu.sh)
  profiles/<tag>_kernel_stats_direct_packer.csv  the same with LACX_DIRECT_PACKER=1 (no copies: the analysis kernel undisturbed)
  profiles/<tag>_bench_line_*.json               the bench lines those two runs printed (HIP events inside the profiled process)
  profiles/<tag>_counters.json                   per kernel: mean of every counter per dispatch (SQ passes, GRBM, FETCH/WRITE)
  profiles/valu.json                             what bench.py's roofline.valu block reads (k_analyze<16,1024>)
  profiles/traffic.json                          what bench.py's roofline.traffic reads

usage: summarize_counters.py <tag> [--workload-tag NAME]   (NAME other than "default" skips valu.json / traffic.json)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import kernel_source_sha256  # noqa: E402  (the hash bench.py checks before it quotes these files)

tag = sys.argv[1]
default = not (len(sys.argv) > 3 and sys.argv[2] == "--workload-tag" and sys.argv[3] != "default")
src = os.path.join(root, "gpurun_out", tag)
prof = os.path.join(root, "profiles")


def short(name):
    n = name.replace("void lacx::", "").replace("lacx::", "")
    n = n.split("(")[0]
    return n.replace("Geo<16, 1024> ", "16,1024").replace("Geo<4, 64> ", "4,64").replace("<<", "<").replace(" >", ">")


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = {k: {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in sorted(d.items())} for k, d in agg.items()}

stats = {}
for name, dst in (("trace", "kernel_stats_bench"), ("trace1", "kernel_stats_direct_packer")):
    for p in glob.glob(os.path.join(src, name, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(p, os.path.join(prof, f"{tag}_{dst}.csv"))
        stats[name] = {short(r["Name"]): r for r in csv.DictReader(open(p))}

for name, dst in (("trace", "bench_line_under_rocprof"), ("trace1", "bench_line_under_rocprof_direct_packer")):
    logp = os.path.join(src, name + ".log")
    if os.path.exists(logp):
        lines = [ln for ln in open(logp) if ln.startswith("{")]
        if lines:
            open(os.path.join(prof, f"{tag}_{dst}.json"), "w").write(lines[-1])

full = next((k for k in counters if k.startswith("k_analyze") and "16,1024" in k), None)
derived = {}
if full and "trace1" in stats and full in stats["trace1"]:
    c = {k: v["mean"] for k, v in counters[full].items()}
    dur_ns = float(stats["trace1"][full]["AverageNs"])
    simds = 256 * 4
    clk_ghz = (c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0) / dur_ns if c.get("GRBM_GUI_ACTIVE") else None
    derived = {
        "kernel": full,
        "single_chunk_duration_ms": dur_ns / 1e6,  # (the LACX_DIRECT_PACKER=1 trace: the kernel without blit copies beside it)
        "valu_wave_insts": c.get("SQ_INSTS_VALU"),
        "salu_wave_insts": c.get("SQ_INSTS_SALU"),
        "effective_clock_ghz_from_GRBM_GUI_ACTIVE": clk_ghz,
        # issue slots: one wave64 VALU instruction holds a SIMD-32 for 2 cycles
        "valu_issue_frac_at_2.4GHz": (c["SQ_INSTS_VALU"] * 2 / (simds * dur_ns * 2.4)) if c.get("SQ_INSTS_VALU") else None,
        "valu_issue_frac_at_effective_clock": (c["SQ_INSTS_VALU"] * 2 / (simds * dur_ns * clk_ghz)) if c.get("SQ_INSTS_VALU") and clk_ghz else None,
        # SQ_ACTIVE_INST_VALU counts quad-cycles in which a wave has a VALU instruction executing (MI355X_MICROARCH.md)
        "valu_busy_frac_at_effective_clock": (c["SQ_ACTIVE_INST_VALU"] * 4 / (simds * dur_ns * clk_ghz)) if c.get("SQ_ACTIVE_INST_VALU") and clk_ghz else None,
        "wave_cycles_share": {k: c[k] / c["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in c and c.get("SQ_WAVE_CYCLES")},
        "lds_bank_conflict_share_of_lds_cycles": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
    }
out = {"tag": tag, "command": "scripts/profile_valu.sh " + tag + " (kernel traces: bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-workloads, as it runs and with "
       "LACX_DIRECT_PACKER=1; counter passes: --steps 3 --warmup 1, one rocprofv3 --pmc run per counter set)",
       "kernel_source_sha256": kernel_source_sha256(),
       "per_kernel_counter_means_per_dispatch": counters, "derived_k_analyze_full": derived}
json.dump(out, open(os.path.join(prof, f"{tag}_counters.json"), "w"), indent=1)
print(json.dumps(derived, indent=1))
for name, st in stats.items():
    print(name)
    for k, r in st.items():
        print(f"   {k:40s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Percentage']}%")

if default and derived:
    c = counters[full]
    json.dump({
        "kernel": "k_analyze<16,1024>",
        "kernel_source_sha256": kernel_source_sha256(),
        "valu_wave_insts_per_step": derived["valu_wave_insts"],
        "salu_wave_insts_per_step": derived["salu_wave_insts"],
        "valu_busy_frac": derived["valu_busy_frac_at_effective_clock"],
        "effective_clock_ghz": derived["effective_clock_ghz_from_GRBM_GUI_ACTIVE"],
        "single_chunk_duration_ms": derived["single_chunk_duration_ms"],
        "source": f"profiles/{tag}_counters.json (rocprofv3 --pmc SQ_INSTS_VALU ... , LACX_PIPE_CHUNKS=1: the whole 10 min stream in one launch)",
    }, open(os.path.join(prof, "valu.json"), "w"), indent=1)
    # traffic of the default pipeline (3 launches per step)
    fk = counters[full]
    if "FETCH_SIZE" in fk and "WRITE_SIZE" in fk and "trace" in stats:
        launches = int(stats["trace"][full]["Calls"]) // 13  # 3 warm-up + 10 timed steps
        # FETCH/WRITE passes ran with the default pipeline, the SQ passes single-chunk: keep only the pipeline's dispatches
        fvals = [v for v in agg[full]["FETCH_SIZE"]]
        wvals = [v for v in agg[full]["WRITE_SIZE"]]
        f, w = sum(fvals) / len(fvals), sum(wvals) / len(wvals)
        blocks = 1758
        json.dump({
            "kernel": "k_analyze<16,1024>",
            "kernel_source_sha256": kernel_source_sha256(),
            "command": "bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end (default workload: interleaved int16 device PCM, device emit), "
                       "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (scripts/profile_valu.sh)",
            "launches_per_step": launches,
            "fetch_size_kb_per_launch": f,
            "write_size_kb_per_launch": w,
            "traffic_bytes_per_launch": int(2 * f * 1024 + w * 1024),
            "algorithmic_bytes_per_launch": int((28_800_000 * 2 * 2 + 2 * blocks * 296) / launches),
            "algorithmic_bytes_incl_bitstream_per_launch": int((28_800_000 * 2 * 2 + 2 * blocks * 296 + 72_573_825) / launches),
            "traffic_ratio": round((2 * f * 1024 + w * 1024) / ((28_800_000 * 2 * 2 + 2 * blocks * 296) / launches), 3),
            "note": "KB -> bytes x1024; FETCH_SIZE x2 (16-byte streaming reads on gfx950, MI355X_MICROARCH.md), WRITE_SIZE as is. "
                    "Algorithmic (SURVEY 8d): PCM at its source depth once + plan records; the fused emit writes the 72.57 MB of "
                    "bitstream once as well (the *_incl_bitstream figure).",
        }, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
