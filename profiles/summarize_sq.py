#!/usr/bin/env python3
"""Summarise rocprofv3 SQ counter passes (--kernel-trace --pmc <SQ_* ...> GRBM_GUI_ACTIVE --output-format csv) per kernel.

    python profiles/summarize_sq.py <out.json> [--windows N] <counter_collection.csv> [<counter_collection.csv> ...]

--windows N: the batch of the window kernels (one workgroup per window): adds config.frames_per_gpu and, per kernel, mfma_instructions_per_window =
SQ_INSTS_VALU_MFMA_F64 / N (bench.py cross-checks its analytic MFMA count against profiles/r*_pmc_mfma.json in this form).

Per kernel: the mean of every counter over its LARGEST dispatches (the top quarter by GRBM_GUI_ACTIVE: the full-batch launches of the timed steps), and
  valu_issue_frac = SQ_INSTS_VALU * 4 / ((GRBM_GUI_ACTIVE / 8) * 1024)     a wave64 VALU instruction occupies its SIMD's issue port for 4 cycles; 1024 SIMDs; GRBM_GUI_ACTIVE
                                                                           is reported summed over the 8 XCDs
  wait_frac       = SQ_WAIT_ANY / SQ_WAVE_CYCLES                           share of the wave cycles spent waiting (s_waitcnt, barriers)
  MfmaUtil        = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 1024)
  mfma_instructions_per_workgroup = SQ_INSTS_VALU_MFMA_F64 * 4 waves... (reported per dispatch; divide by the windows of the batch)"""
import collections, csv, json, sys


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    windows = None
    if paths and paths[0] == "--windows":
        windows, paths = int(paths[1]), paths[2:]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            vals[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"source": "rocprofv3 --kernel-trace --pmc <counters> GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py ... (tools/dev_pmc_pipes.sh, tools/gpu_round.sh)", "note": __doc__.split("Per kernel:")[1].strip(), "kernels": {}}
    for k, d in sorted(vals.items()):
        def top(n):
            v = sorted(d.get(n, [0.0])); v = v[len(v) * 3 // 4:] or [0.0]
            return sum(v) / len(v)
        g = top("GRBM_GUI_ACTIVE")
        if g < 8 * 24000:            # shorter than ~10 us: not a kernel of the steady state
            continue
        e = {n: top(n) for n in sorted(d)}
        e["dispatches"] = len(d.get("GRBM_GUI_ACTIVE", []))
        simd_cycles = g / 8 * 1024
        if "SQ_INSTS_VALU" in d: e["valu_issue_frac"] = e["SQ_INSTS_VALU"] * 4 / simd_cycles
        if "SQ_INSTS_SALU" in d: e["salu_issue_frac"] = e["SQ_INSTS_SALU"] / simd_cycles
        if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d and e["SQ_WAVE_CYCLES"] > 0: e["wait_frac"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d: e["MfmaUtil_percent"] = 100.0 * e["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
        if windows and "SQ_INSTS_VALU_MFMA_F64" in d: e["mfma_instructions_per_window"] = e["SQ_INSTS_VALU_MFMA_F64"] / windows; e["mfma_flop_per_dispatch"] = e["SQ_INSTS_VALU_MFMA_F64"] * 2048.0
        if "SQ_LDS_BANK_CONFLICT" in d and e.get("SQ_LDS_IDX_ACTIVE", 0) > 0: e["lds_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
        res["kernels"][k] = e
    if windows: res["config"] = {"frames_per_gpu": windows}
    json.dump(res, open(out, "w"), indent=1)
    for k, e in res["kernels"].items():
        print("%-44s VALU %5.1f %%  wait %5.1f %%  MFMA %5.1f %%" % (k[:44], 100 * e.get("valu_issue_frac", 0), 100 * e.get("wait_frac", 0), e.get("MfmaUtil_percent", 0)))


if __name__ == "__main__":
    main()
