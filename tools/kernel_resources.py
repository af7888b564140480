#!/usr/bin/env python3
"""Registers, spills, LDS and scratch of every kernel in libvilfusion_hip.so, read from the gfx950 code objects' metadata notes
(the same numbers `llvm-readelf --notes` shows). Writes vil_fusion_amd/csrc/kernel_resources.json; bench.py quotes the window kernels from it.

  python tools/kernel_resources.py [pattern]      # prints name, vgpr, agpr, sgpr, spills, scratch B/lane, static LDS B, waves/SIMD by registers
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "vil_fusion_amd", "csrc", "libvilfusion_hip.so")
OUT = os.path.join(ROOT, "vil_fusion_amd", "csrc", "kernel_resources.json")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """the gfx950 ELF images inside the host library's clang offload bundles"""
    data = open(path, "rb").read()
    for m in re.finditer(MAGIC, data):
        base = m.start()
        n, = struct.unpack_from("<Q", data, base + len(MAGIC))
        off = base + len(MAGIC) + 8
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24: off + 24 + tl].decode()
            off += 24 + tl
            if "gfx950" in triple and size > 0:
                yield data[base + o: base + o + size]


def waves_per_simd(alloc):
    """MI355X_MICROARCH.md, Register files: granule 8, 512 per SIMD lane"""
    a = (alloc + 7) // 8 * 8
    return min(8, 512 // max(a, 1))


def read(path=SO):
    kernels = {}
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for block in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            block = ".agpr_count:" + block
            g = lambda key, d=0: (lambda m_: int(m_.group(1)) if m_ else d)(re.search(r"\." + key + r":\s+(\d+)", block))
            nm = re.search(r"\.name:\s+(\S+)", block)
            if not nm:
                continue
            v, a = g("vgpr_count"), g("agpr_count")
            kernels[nm.group(1)] = dict(vgpr=v, agpr=a, sgpr=g("sgpr_count"), vgpr_spills=g("vgpr_spill_count"), sgpr_spills=g("sgpr_spill_count"),
                                        scratch_bytes_per_lane=g("private_segment_fixed_size"), static_lds_bytes=g("group_segment_fixed_size"),
                                        max_flat_workgroup_size=g("max_flat_workgroup_size"), waves_per_simd_by_registers=waves_per_simd(v))
    return kernels


def main():
    k = read()
    json.dump(k, open(OUT, "w"), indent=1, sort_keys=True)
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for name in sorted(k):
        if pat in name:
            r = k[name]
            print(f"{name:44s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} spills {r['vgpr_spills']:3d} scratch {r['scratch_bytes_per_lane']:4d} B  lds {r['static_lds_bytes']:6d} B  "
                  f"waves/SIMD {r['waves_per_simd_by_registers']}")


if __name__ == "__main__":
    main()
