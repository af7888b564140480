"""The drop-in boundary: libvilfusion_hip.so loads without a GPU, exports every function include/vilfusion.h declares, and the
ctypes mirrors in vil_fusion_amd/abi.py have exactly the C layouts (checked by compiling the header with gcc)."""
import ctypes as C
import os
import re
import subprocess
import sys
from vil_fusion_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vilfusion.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vilf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_function():
    names = _declared_functions()
    assert len(names) > 40 and "vilf_window_solve" in names and "vilf_scan2map_batch_step" in names
    so = os.path.join(ROOT, "vil_fusion_amd", "csrc", "libvilfusion_hip.so")
    if not os.path.exists(so):
        lib.build()
    L = C.CDLL(so)                                    # must load on a machine without a GPU (no compute call is made here)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(names) <= set(lib.EXPORTED) | {"vilf_handle"}, sorted(set(names) - set(lib.EXPORTED))
    L.vilf_version.restype = C.c_char_p
    assert L.vilf_version().decode().startswith("vilfusion")


def test_ctypes_mirrors_match_the_c_layout(tmp_path):
    structs = {"vilf_options": abi.Options, "vilf_imu_preint": abi.ImuPreint, "vilf_lidar_constraint": abi.LidarConstraint, "vilf_window_in": abi.WindowIn,
               "vilf_summary": abi.Summary, "vilf_window_out": abi.WindowOut, "vilf_prior": abi.Prior, "vilf_imu_noise": abi.ImuNoise,
               "vilf_scan2map_result": abi.Scan2MapResult}
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "vilfusion.h"', "int main(void) {"]
    checks = []
    for cname, ct in structs.items():
        prog.append(f'  printf("%zu\\n", sizeof({cname}));')
        checks.append((cname, "sizeof", C.sizeof(ct)))
        for fname, _ in ct._fields_:
            prog.append(f'  printf("%zu\\n", offsetof({cname}, {fname}));')
            checks.append((cname, fname, getattr(ct, fname).offset))
    prog += ["  return 0;", "}"]
    src = tmp_path / "layout.c"; exe = tmp_path / "layout"
    src.write_text("\n".join(prog))
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert len(out) == len(checks)
    bad = [(c, f, int(o), e) for (c, f, e), o in zip(checks, out) if int(o) != e]
    assert not bad, bad
