#!/usr/bin/env python3
"""Summarise a rocprofv3 MFMA counter pass of the solve-only bench (tools/gpu_round.sh: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE, --kernel-trace --output-format csv) per back-end kernel.

    python profiles/summarize_pmc_mfma.py <counter_collection.csv> <out json> [frames_per_gpu]

SQ_INSTS_VALU_MFMA_F64 counts MFMA instructions per wave (one v_mfma_f64_16x16x4_f64 = 2048 flop = 64 busy cycles of its SIMD). GRBM_GUI_ACTIVE is reported
summed over the 8 XCDs: MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 1024 SIMDs), the rocprofv3 MfmaUtil expression."""
import collections
import csv
import json
import sys


def main():
    path, out_path = sys.argv[1], sys.argv[2]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        # one workgroup per window: only the dispatches over the full batch count (the bench's PCIe-inclusive leg solves a smaller batch with the same kernels)
        if k in ("k_linearize", "k_linearize_last", "k_solve", "k_solve_sb", "k_step") and int(r["Grid_Size"]) == frames * int(r["Workgroup_Size"]):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE "
                     "--output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lidar-stage --no-marginalize --ragged-windows 0 (tools/gpu_round.sh)",
           "note": __doc__.split("SQ_INSTS_VALU_MFMA_F64 counts")[1].strip(), "config": {"frames_per_gpu": frames}, "kernels": {}}
    for k, d in acc.items():
        e = {c: sum(v) / max(len(v), 1) for c, v in d.items()}
        e["dispatches"] = len(d.get("SQ_INSTS_VALU_MFMA_F64", []))
        if e.get("GRBM_GUI_ACTIVE"):
            e["MfmaUtil_percent"] = 100.0 * e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / ((e["GRBM_GUI_ACTIVE"] / 8.0) * 1024.0)
        e["mfma_instructions_per_window"] = e.get("SQ_INSTS_VALU_MFMA_F64", 0.0) / frames
        e["mfma_flop_per_dispatch"] = e.get("SQ_INSTS_VALU_MFMA_F64", 0.0) * 2048.0
        out["kernels"][k] = e
    json.dump(out, open(out_path, "w"), indent=1)
    for k, e in out["kernels"].items():
        print(k, "dispatches", e["dispatches"], "MFMA instr / window", round(e["mfma_instructions_per_window"], 1), "MfmaUtil %", round(e.get("MfmaUtil_percent", 0.0), 2))


if __name__ == "__main__":
    main()
