"""Build libntracer_hip.so in-tree:  python -m ntracer_amd.build

hipcc cross-compiles gfx950 code objects without a GPU.  The kernels are templates over the dimension; every
dimension is its own translation unit (csrc/nt_inst_box.hip with -DNT_INST_N=3..24 and csrc/nt_inst_composite.hip with 3..10),
compiled in parallel into build/*.o and linked with the host side.  -ffp-contract=off is part of the arithmetic
contract with the oracle (see csrc/nt_pixel.hpp); -fno-slp-vectorize because packing pairs of independent fp32
operations into v_pk_* costs more register shuffling than it saves here (measured: 2-4 %)."""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libntracer_hip.so")
DIMS = range(3, 11)
BOX_ONLY_DIMS = range(11, 25)         # BoxScene kernels alone are also compiled for N = 11..24
HDR = [os.path.join(CSRC, h) for h in ("nt_device.hpp", "nt_pixel.hpp", "nt_box.hpp", "nt_composite.hpp")] + \
      [os.path.join(HERE, "..", "include", "ntracer_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-pthread", "-fno-slp-vectorize", "-Wall",
         "-Wno-unused-function"]
EXTRA = os.environ.get("NTRACER_HIPCC_FLAGS", "").split()         # ablation builds (-DNT_EXP_...)


def units():
    """(object name, source, extra flags)"""
    u = [("nt_api", "nt_api.cpp", []), ("nt_builder", "nt_builder.cpp", []), ("nt_launch", "nt_launch.cpp", []),
         ("nt_var", "nt_var.hip", [])]
    for n in DIMS:
        u.append(("nt_box_%d" % n, "nt_inst_box.hip", ["-DNT_INST_N=%d" % n]))
        u.append(("nt_composite_%d" % n, "nt_inst_composite.hip", ["-DNT_INST_N=%d" % n]))
    for n in BOX_ONLY_DIMS:
        u.append(("nt_box_%d" % n, "nt_inst_box.hip", ["-DNT_INST_N=%d" % n]))
    return u


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps)


def _flag_tag():
    return hashlib.sha1(" ".join(FLAGS + EXTRA).encode()).hexdigest()[:8]


PYBUF_SRC = os.path.join(CSRC, "nt_pybuffer.c")
# named with the interpreter's ABI tag (_pybuffer.cpython-310-x86_64-linux-gnu.so): a prebuilt copy that travels to a box
# with another CPython is then simply not found -- render.py falls back to `object` -- instead of being dlopened against
# the wrong PyTypeObject layout
import sysconfig as _sysconfig
PYBUF_OUT = os.path.join(HERE, "_pybuffer" + (_sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_pybuffer(force=False, verbose=False):
    """the one-type CPython module that gives Vector / Color the buffer protocol (host convenience; plain gcc)"""
    if not force and not _stale(PYBUF_OUT, [PYBUF_SRC]):
        return PYBUF_OUT
    import sysconfig
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-shared", "-I" + sysconfig.get_paths()["include"], PYBUF_SRC, "-o", PYBUF_OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(PYBUF_OUT + ".tmp", PYBUF_OUT)
    return PYBUF_OUT


def build(force=False, verbose=False, out=None):
    if out is None:
        build_pybuffer(force, verbose)
    out = out or OUT
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    tag = _flag_tag()
    jobs = []
    objs = []
    for name, src, extra in units():
        o = os.path.join(OBJ, "%s.%s.o" % (name, tag))
        objs.append(o)
        s = os.path.join(CSRC, src)
        if force or _stale(o, [s] + HDR):
            jobs.append([cc] + FLAGS + EXTRA + extra + ["-c", s, "-o", o])
    if jobs:
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        # the composite units are the long ones (~40 s each): start them first
        jobs.sort(key=lambda c: 0 if "nt_inst_composite.hip" in c[-3] else 1)
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    # the flags the library was last linked from, beside it: a default build after an ablation build (NTRACER_HIPCC_FLAGS) finds
    # its own objects up to date, and would otherwise leave the ablation library in place
    tagfile = out + ".tag"
    try:
        linked_tag = open(tagfile).read().strip()
    except OSError:
        linked_tag = ""
    if jobs or force or linked_tag != tag or _stale(out, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", out + ".tmp"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(out + ".tmp", out)
        with open(tagfile, "w") as fh:
            fh.write(tag + "\n")
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    print(build(force="--force" in sys.argv, verbose=True, out=args[0] if args else None))
