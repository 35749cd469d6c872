#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of every kernel in the built objects: python3 tools/kernel_resources.py [substring]
(reads the gfx950 code objects out of ntracer_amd/build/*.o with clang-offload-bundler + llvm-readelf --notes)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
want = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
for o in sorted(glob.glob(os.path.join(ROOT, "ntracer_amd", "build", "*.o"))):
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "dev.co")
        fat = os.path.join(td, "fat.bin")
        r = subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, o], capture_output=True)
        if r.returncode or not os.path.exists(fat):
            continue
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--output=" + co], capture_output=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count")[1:]:
        def f(k):
            m = re.search(r"\.%s:\s+(\S+)" % k, blk)
            return m.group(1) if m else "?"
        name = f("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem)
        dem = re.sub(r"\(.*", "", dem).replace("void ", "")
        if want in dem:
            rows.append((dem, f("vgpr_count"), f("sgpr_count"), f("private_segment_fixed_size"), f("group_segment_fixed_size")))
print("%-60s %5s %5s %8s %6s" % ("kernel", "vgpr", "sgpr", "scratch", "lds"))
for r in sorted(set(rows)):
    print("%-60s %5s %5s %8s %6s" % r)
