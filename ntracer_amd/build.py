"""Build libntracer_hip.so in-tree:  python -m ntracer_amd.build

hipcc cross-compiles gfx950 code objects without a GPU.  -ffp-contract=off is part of the
arithmetic contract with the oracle (see csrc/nt_kernels.hip); -fno-slp-vectorize because packing pairs of
independent fp32 operations into v_pk_* costs more register shuffling than it saves here (measured: 2-4 %)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", "nt_api.cpp"), os.path.join(HERE, "csrc", "nt_builder.cpp"), os.path.join(HERE, "csrc", "nt_kernels.hip")]
HDR = [os.path.join(HERE, "csrc", "nt_device.hpp"), os.path.join(HERE, "..", "include", "ntracer_hip.h")]
OUT = os.path.join(HERE, "libntracer_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-fno-slp-vectorize", "-Wall",
         "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(p) <= t for p in SRC + HDR)


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    cmd = [hipcc()] + FLAGS + SRC + ["-o", OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
