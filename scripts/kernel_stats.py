#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel of a translation unit, from the assembly hipcc emits.

    python scripts/kernel_stats.py tactics2d_amd/csrc/t2d_collide.hip [filter] [-Dflags...]

Prints one line per kernel: private segment (scratch) bytes, SGPRs, spilled SGPRs, VGPRs, spilled VGPRs, static LDS and --
from the body -- static instruction counts by class (VALU 64-bit / 32-bit / packed, SALU, LDS, VMEM, branches, waits)."""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
flt = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
extra = [a for a in sys.argv[2:] if a.startswith("-")]
out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
       "--cuda-device-only", "-S", src, "-o", out] + extra
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


meta = {}
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
    name = g("name").group(1)
    meta[name] = dict(priv=int(g("private_segment_fixed_size").group(1)), sgpr=int(g("sgpr_count").group(1)),
                      sspill=int(g("sgpr_spill_count").group(1)), vgpr=int(g("vgpr_count").group(1)),
                      vspill=int(g("vgpr_spill_count").group(1)), lds=int(g("group_segment_fixed_size").group(1)))
for name, m in meta.items():
    dem = demangle(name)
    if flt and flt not in dem:
        continue
    body = re.search(r"^%s:.*?\n(.*?)\n\.Lfunc_end" % re.escape(name), txt, re.S | re.M)
    cls = dict(valu64=0, valu32=0, pk=0, salu=0, lds=0, vmem=0, branch=0, wait=0)
    if body:
        for line in body.group(1).splitlines():
            t = line.strip().split()
            if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
                continue
            op = t[0]
            if op.startswith("v_pk_"):
                cls["pk"] += 1
            elif op.startswith("v_"):
                cls["valu64" if re.search(r"(f64|b64|u64|i64)", op) else "valu32"] += 1
            elif op.startswith(("s_cbranch", "s_branch")):
                cls["branch"] += 1
            elif op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
                cls["wait"] += 1
            elif op.startswith("s_"):
                cls["salu"] += 1
            elif op.startswith("ds_"):
                cls["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                cls["vmem"] += 1
    short = dem[dem.find("::", dem.find("anonymous")) + 2:] if "anonymous" in dem else dem
    short = short[:short.find(">(") + 1] if ">(" in short else short[:80]
    print(f"{short:78s} priv {m['priv']:4d} sgpr {m['sgpr']:3d} sspill {m['sspill']:3d} vgpr {m['vgpr']:3d} vspill {m['vspill']:3d} "
          f"lds {m['lds']:6d} | " + " ".join(f"{k} {v}" for k, v in cls.items()))
