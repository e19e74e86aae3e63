"""Build libt2d_hip.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m tactics2d_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract
(DESIGN.md "Precision"): every product and sum rounds separately unless the source says fma.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, os.environ.get("T2D_LIB_NAME", "libt2d_hip.so"))
SOURCES = ["t2d_api.hip", "t2d_integrate.hip", "t2d_collide.hip", "t2d_lidar.hip", "t2d_idm.hip", "t2d_drift.hip",
           "t2d_generate.hip", "t2d_ego.hip", "t2d_frame.hip", "t2d_mapgrid.hip", "t2d_geometry_host.hip"]
# test / measurement hooks (include/t2d_debug.h): compiled into libt2d_hip_debug.so only
DEBUG_SOURCES = ["t2d_loop.hip"]
HEADERS = ["t2d_math.h", "t2d_pool.h", "t2d_host.h", "t2d_integrate_dev.h", "t2d_geom_dev.h", "t2d_idm_dev.h", "t2d_scene_dev.h",
           os.path.join("..", "..", "include", "t2d.h")]
DEBUG_HEADERS = [os.path.join("..", "..", "include", "t2d_debug.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
          "-fno-fast-math", "-Wall", "-Wno-unused-function", "-pthread"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"]
FLAGS = CFLAGS + ["-shared"]   # (what source_hash covers: the flags every translation unit is compiled with)
# flags of single translation units (part of the source hash)
FILE_FLAGS = {}


def source_hash():
    """sha256 over the kernel sources, headers and build flags: what a counter pass (profiles/traffic_latest.json) was taken
    of.  bench.py compares it with the tree it runs from and marks the instruction counts stale when they differ."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(FILE_FLAGS):
        h.update((f + " " + " ".join(FILE_FLAGS[f])).encode())
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def _deps(debug=False):
    return [os.path.join(CSRC, f) for f in SOURCES + HEADERS + (DEBUG_SOURCES + DEBUG_HEADERS if debug else [])] + [os.path.abspath(__file__)]


def _stale(lib, debug=False):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps(debug))


def needs_build():
    return _stale(LIB)


def _compile_link(lib, sources, extra, verbose=False):
    """one hipcc per translation unit, side by side, then the link: a build is as long as its longest file (t2d_collide.hip)"""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "_obj", os.path.splitext(os.path.basename(lib))[0])
    os.makedirs(objdir, exist_ok=True)

    def one(f):
        obj = os.path.join(objdir, os.path.splitext(f)[0] + ".o")
        cmd = [hipcc] + CFLAGS + FILE_FLAGS.get(f, []) + extra + ["-c", os.path.join(CSRC, f), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(int(os.environ.get("T2D_BUILD_JOBS", "6"))) as ex:
        objs = list(ex.map(one, sources))
    tmp = lib + ".tmp"
    subprocess.check_call([hipcc] + LDFLAGS + objs + ["-ldl", "-o", tmp])
    os.replace(tmp, lib)
    return lib


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    extra = os.environ.get("T2D_EXTRA_FLAGS", "").split()
    return _compile_link(LIB, SOURCES, extra, verbose)


# the conservative build tests/test_gpu_soak.py holds the product against (see wave_sync in t2d_collide.hip)
CHECK_LIB = os.path.join(HERE, "libt2d_hip_waitcnt.so")
CHECK_FLAGS = ["-DT2D_WAVE_SYNC_WAITCNT"]


def build_check_lib(force=False):
    """libt2d_hip_waitcnt.so: the same sources with every wave-level LDS sync preceded by s_waitcnt lgkmcnt(0)."""
    if not force and not _stale(CHECK_LIB):
        return CHECK_LIB
    return _compile_link(CHECK_LIB, SOURCES, CHECK_FLAGS)


# the product's sources + the test / measurement hooks of include/t2d_debug.h (fault injection, a gather delay, a stand-in
# policy with its closed-loop runner, placement maps): what tests/, bench.py's closed_loop leg and scripts/ load through
# tactics2d_amd/debug.py.  The product library exports none of it (tests/test_layout.py).
DEBUG_LIB = os.path.join(HERE, "libt2d_hip_debug.so")
DEBUG_FLAGS = ["-DT2D_DEBUG_HOOKS"]


def build_debug_lib(force=False):
    if not force and not _stale(DEBUG_LIB, debug=True):
        return DEBUG_LIB
    return _compile_link(DEBUG_LIB, SOURCES + DEBUG_SOURCES, DEBUG_FLAGS)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--check-lib" in sys.argv:
        print(build_check_lib(force="--force" in sys.argv))
    if "--debug-lib" in sys.argv:
        print(build_debug_lib(force="--force" in sys.argv))
