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
           "t2d_generate.hip", "t2d_ego.hip", "t2d_loop.hip", "t2d_frame.hip"]
HEADERS = ["t2d_math.h", "t2d_pool.h", "t2d_integrate_dev.h", "t2d_geom_dev.h", "t2d_idm_dev.h", "t2d_scene_dev.h",
           os.path.join("..", "..", "include", "t2d.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-pthread"]


def source_hash():
    """sha256 over the kernel sources, headers and build flags: what a counter pass (profiles/traffic_latest.json) was taken
    of.  bench.py compares it with the tree it runs from and marks the instruction counts stale when they differ."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("T2D_EXTRA_FLAGS", "").split()
    cmd = [hipcc] + FLAGS + extra + [os.path.join(CSRC, f) for f in SOURCES] + ["-ldl", "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


# the conservative build tests/test_gpu_soak.py holds the product against (see wave_sync in t2d_collide.hip)
CHECK_LIB = os.path.join(HERE, "libt2d_hip_waitcnt.so")
CHECK_FLAGS = ["-DT2D_WAVE_SYNC_WAITCNT"]


def build_check_lib(force=False):
    """libt2d_hip_waitcnt.so: the same sources with every wave-level LDS sync preceded by s_waitcnt lgkmcnt(0)."""
    if not force and os.path.exists(CHECK_LIB):
        t = os.path.getmtime(CHECK_LIB)
        deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
        if all(os.path.getmtime(d) <= t for d in deps):
            return CHECK_LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc] + FLAGS + CHECK_FLAGS + [os.path.join(CSRC, f) for f in SOURCES] + ["-ldl", "-o", CHECK_LIB])
    return CHECK_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--check-lib" in sys.argv:
        print(build_check_lib(force="--force" in sys.argv))
