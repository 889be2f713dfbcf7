#!/usr/bin/env python3
"""Kernel metadata of the gfx950 code object inside a built library (registers, spills, scratch, LDS):
    python scripts/kernel_meta.py [lib.so] [name-filter]
Unbundles the code object (clang-offload-bundler) and reads the AMDGPU metadata note (llvm-readelf --notes)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_meta(lib):
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "co"); fb = os.path.join(td, "fb.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
        txt = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    out = []
    for blk in txt.split("- .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
        out.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), spill=g("vgpr_spill_count"),
                        scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size")))
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "thinshelllab_amd", "lib", "libtsl_hip.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in sorted(kernel_meta(lib), key=lambda k: k["name"]):
        if flt in k["name"]:
            dem = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
            print(f"{dem[:70]:70s} vgpr {k['vgpr']:>4} agpr {k['agpr']:>4} spill {k['spill']:>5} scratch {k['scratch']:>5} lds {k['lds']:>6}")
