#!/usr/bin/env python3
"""Registers / spills / LDS of the gfx950 kernels in a built object file, from the code object's metadata notes.
    python tools/kernel_resources.py [gp-ann_amd/build/scan.o] [name fragment ...]"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else "gp-ann_amd/build/scan.o"
    frags = [a for a in sys.argv[1:] if not os.path.exists(a)]
    with tempfile.TemporaryDirectory() as tmp:
        co, fat = os.path.join(tmp, "k.co"), os.path.join(tmp, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    rows = []
    for blk in notes.split("- .agpr_count:")[1:]:
        def g(key):
            m = re.search(rf"\.{key}:\s+(\S+)", blk)
            return m.group(1) if m else "?"
        name = g("name")
        try:
            name = subprocess.check_output([f"{LLVM}/llvm-cxxfilt", name], text=True).strip()
        except Exception:
            pass
        if frags and not all(f in name for f in frags):
            continue
        rows.append((name, g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"),
                     g("group_segment_fixed_size")))
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>7}  kernel")
    for name, v, a, sg, vs, ss, sc, lds in sorted(rows):
        print(f"{v:>5} {a:>5} {sg:>5} {vs:>6} {ss:>6} {sc:>7} {lds:>7}  {name[:150]}")


if __name__ == "__main__":
    main()
