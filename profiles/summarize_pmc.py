#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes of bench.py (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace
--output-format csv) into per-kernel and per-launch-group HBM traffic.

    python profiles/summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <steps incl. warm-up> <out prefix> [frames_per_gpu] [workload_tag]

Counter unit: KB per dispatch. gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly half the bytes of a WIDE COALESCED streaming read (16 B per
lane), so corrected bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; raw = (FETCH_SIZE + WRITE_SIZE) * 1024. The guide calibrates the factor for that access shape only:
for the gather kernels (GATHER below: the 5-NN walk, the directory build, the LM solve's factor records, the window kernels' 8-byte operands) "corrected" is an UPPER
bound and the truth lies between the two figures — both are written side by side. Groups are bench.py's launch groups (vilf_get_profile*): traffic per group launch =
group bytes per step / group launches per step."""
import collections
import csv
import json
import sys

GROUPS = {   # kernel-name prefix -> bench.py launch group
    "k_linearize_last": "k_step", "k_linearize": "k_linearize", "k_solve": "k_solve", "k_step": "k_step",      # k_linearize_last = the step-only launch that ends a solve
    "k_marg_prepare": "k_marg_prepare", "k_marg_schur": "k_marg_schur", "k_marg_finish": "k_marg_finish", "k_mf_": "k_marg_finish", "k_prior_prep": "k_prior_prep",
    "b_minmax": "s2m_voxel_grid", "b_voxel_keys": "s2m_voxel_grid", "void b_voxel_keys": "s2m_voxel_grid", "void b_voxel_reduce": "s2m_voxel_grid",
    "void b_voxel_heads": "s2m_voxel_grid", "void b_map_update": "s2m_map_update",
    "b_check_order": "s2m_voxel_grid", "void b_scan_voxel": "s2m_voxel_grid", "b_scan_voxel_runs": "s2m_voxel_grid",
    "void rocprim": "s2m_radix_sort", "b_bucket_index": "s2m_neighbour_index", "b_dir_build": "s2m_neighbour_index", "b_gather_sorted": "s2m_neighbour_index",
    "b_make_cid": "s2m_neighbour_index", "b_associate": "s2m_associate", "b_solve": "s2m_lm_solve",
    "b_crop_compact": "s2m_submap", "b_transform_append": "s2m_submap", "b_bump": "s2m_submap",
}
LAUNCHES_PER_STEP = {"k_linearize": 8, "k_solve": 8, "k_step": 1, "k_marg_prepare": 1, "k_marg_schur": 1, "k_marg_finish": 1, "k_prior_prep": 1,
                     "s2m_voxel_grid": 2, "s2m_map_update": 2, "s2m_radix_sort": 1, "s2m_neighbour_index": 2, "s2m_associate": 2, "s2m_lm_solve": 2, "s2m_submap": 2}
GATHER = ("s2m_associate", "s2m_neighbour_index", "s2m_lm_solve", "k_linearize", "k_solve", "k_step", "k_marg_prepare", "k_marg_schur", "k_marg_finish", "k_prior_prep")


def group_of(name):
    for k, g in GROUPS.items():
        if name.startswith(k):
            return g
    return None


def main():
    fetch_csv, write_csv, steps, prefix = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    frames = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
    tag = sys.argv[6] if len(sys.argv) > 6 else "lidar+solve+marginalize"
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    for cname, path in (("FETCH_SIZE", fetch_csv), ("WRITE_SIZE", write_csv)):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == cname:
                kn = r["Kernel_Name"].split("(")[0][:80]
                # the back-end kernels run one workgroup per window: only their dispatches over the full batch count (the bench's PCIe-inclusive leg and the ragged
                # line solve other batch sizes with the same kernels, outside the timed steps)
                if kn in ("k_linearize", "k_linearize_last", "k_solve", "k_solve_sb", "k_step") and int(r["Grid_Size"]) != frames * int(r["Workgroup_Size"]):
                    continue
                per_kernel[kn][cname].append(float(r["Counter_Value"]))
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- python3 bench.py --steps S --warmup W --no-cpu-baseline (two passes)",
           "note": __doc__.split("Counter unit:")[1].strip(), "steps_in_run": steps, "config": {"frames_per_gpu": frames, "workload_tag": tag}, "kernels": {}, "groups": {}}
    rows = []
    groups = collections.defaultdict(lambda: [0.0, 0.0])
    for k, d in sorted(per_kernel.items()):
        f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
        fs, ws = sum(f), sum(w)
        rows.append((k, len(f), fs / max(len(f), 1), ws / max(len(w), 1)))
        g = group_of(k)
        if g:
            groups[g][0] += fs; groups[g][1] += ws
        if k.startswith(("k_", "b_", "void b_")):
            out["kernels"][k] = {"dispatches": len(f), "FETCH_SIZE_KB_mean": fs / max(len(f), 1), "WRITE_SIZE_KB_mean": ws / max(len(w), 1),
                                 "hbm_bytes_per_launch_raw": (fs / max(len(f), 1) + ws / max(len(w), 1)) * 1024,
                                 "hbm_bytes_per_launch_corrected": (2 * fs / max(len(f), 1) + ws / max(len(w), 1)) * 1024}
    for g, (fs, ws) in groups.items():
        lps = LAUNCHES_PER_STEP.get(g, 1)
        out["groups"][g] = {"hbm_bytes_per_step_corrected": (2 * fs + ws) * 1024 / steps, "hbm_bytes_per_step_raw": (fs + ws) * 1024 / steps,
                            "launches_per_step": lps, "hbm_bytes_per_launch_corrected": (2 * fs + ws) * 1024 / steps / lps,
                            "hbm_bytes_per_launch_raw": (fs + ws) * 1024 / steps / lps,
                            "access": "gather / narrow: corrected is an upper bound" if g in GATHER else "wide coalesced streams: corrected applies"}
    json.dump(out, open(prefix + "_pmc_traffic.json", "w"), indent=1)
    with open(prefix + "_pmc_summary.csv", "w") as fh:
        fh.write("kernel,dispatches,mean_FETCH_SIZE_KB,mean_WRITE_SIZE_KB\n")
        for r in rows:
            fh.write('"%s",%d,%.3f,%.3f\n' % r)
    print(json.dumps(out["groups"], indent=1))


if __name__ == "__main__":
    main()
