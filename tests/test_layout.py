"""The Python-side constants and the ctypes symbol table must agree with include/t2d.h, and the
built shared library must export every symbol the header declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "t2d.h")).read()


def _enum_values():
    vals = {}
    for name, v in re.findall(r"#define\s+(T2D_\w+)\s+(\d+)u?\b", HEADER):
        vals[name] = int(v)
    for body in re.findall(r"enum\s*\{(.*?)\};", HEADER, re.S):
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        for name, v in re.findall(r"(T2D_\w+)\s*=\s*(\d+)", body):
            vals[name] = int(v)
    return vals


def test_layout_constants_match_header():
    from tactics2d_amd import layout as L
    h = _enum_values()
    pairs = {"P_MODEL": "T2D_P_MODEL", "P_LF": "T2D_P_LF", "P_LR": "T2D_P_LR", "P_WB": "T2D_P_WB",
             "P_STEER_LO": "T2D_P_STEER_LO", "P_STEER_HI": "T2D_P_STEER_HI", "P_SPEED_LO": "T2D_P_SPEED_LO",
             "P_SPEED_HI": "T2D_P_SPEED_HI", "P_ACCEL_LO": "T2D_P_ACCEL_LO", "P_ACCEL_HI": "T2D_P_ACCEL_HI",
             "P_RANGE_FLAGS": "T2D_P_RANGE_FLAGS", "P_MASS": "T2D_P_MASS", "P_MASS_HEIGHT": "T2D_P_MASS_HEIGHT",
             "P_MU": "T2D_P_MU", "P_IZ": "T2D_P_IZ", "P_CF": "T2D_P_CF", "P_CR": "T2D_P_CR",
             "P_DELTA_T_MS": "T2D_P_DELTA_T_MS", "P_SHAPE": "T2D_P_SHAPE", "P_LENGTH": "T2D_P_LENGTH",
             "P_WIDTH": "T2D_P_WIDTH", "PARAM_COLS": "T2D_PARAM_COLS", "MAX_TYPES": "T2D_MAX_TYPES",
             "RANGE_STEER": "T2D_RANGE_STEER", "RANGE_SPEED": "T2D_RANGE_SPEED", "RANGE_ACCEL": "T2D_RANGE_ACCEL",
             "MODEL_KINEMATICS": "T2D_MODEL_KINEMATICS", "MODEL_DYNAMICS": "T2D_MODEL_DYNAMICS",
             "MODEL_POINTMASS": "T2D_MODEL_POINTMASS", "SHAPE_OBB": "T2D_SHAPE_OBB", "SHAPE_CIRCLE": "T2D_SHAPE_CIRCLE",
             "F_X": "T2D_F_X", "F_Y": "T2D_F_Y", "F_HEADING": "T2D_F_HEADING", "F_SPEED": "T2D_F_SPEED",
             "F_VX": "T2D_F_VX", "F_VY": "T2D_F_VY", "F_ACT0": "T2D_F_ACT0", "F_ACT1": "T2D_F_ACT1",
             "F_IDS": "T2D_F_IDS", "F_FLAGS": "T2D_F_FLAGS", "F_APPLIED0": "T2D_F_APPLIED0",
             "F_APPLIED1": "T2D_F_APPLIED1", "F_ENV_FLAGS": "T2D_F_ENV_FLAGS", "F_CNT_STEP": "T2D_F_CNT_STEP",
             "F_FRAME_MS": "T2D_F_FRAME_MS", "F_STATUS": "T2D_F_STATUS", "F_REWARD": "T2D_F_REWARD",
             "F_COUNT": "T2D_F_COUNT", "FLAG_COLLISION_DYNAMIC": "T2D_FLAG_COLLISION_DYNAMIC",
             "FLAG_COLLISION_STATIC": "T2D_FLAG_COLLISION_STATIC", "FLAG_OUT_BOUND": "T2D_FLAG_OUT_BOUND",
             "FLAG_OFF_LANE": "T2D_FLAG_OFF_LANE", "MAX_POLY_VERTS": "T2D_MAX_POLY_VERTS",
             "MAX_AGENTS": "T2D_MAX_AGENTS"}
    for py, c in pairs.items():
        assert getattr(L, py) == h[c], (py, c)


def test_golden_generator_uses_the_same_columns():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(ROOT, "oracle", "gen_golden.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    from tactics2d_amd import layout as L
    for n in ("P_MODEL", "P_LF", "P_LR", "P_WB", "P_STEER_LO", "P_SPEED_HI", "P_ACCEL_LO", "P_RANGE_FLAGS",
              "P_MASS", "P_MASS_HEIGHT", "P_MU", "P_IZ", "P_CF", "P_CR", "P_SHAPE", "P_LENGTH", "P_WIDTH"):
        assert getattr(g, n) == getattr(L, n)
    assert g.P_DELTA_T == L.P_DELTA_T_MS and g.NCOL == L.PARAM_COLS


DEBUG_HEADER = open(os.path.join(ROOT, "include", "t2d_debug.h")).read()


def _declared_functions(header=HEADER):
    text = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    return sorted(set(re.findall(r"\b(t2d_[a-z_0-9]+)\s*\(", text)) - {"t2d_pool", "t2d_status_config"})


def _exported(path):
    """dynamic symbols a shared library defines (nm -D --defined-only)"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


def test_ffi_table_covers_the_header():
    from tactics2d_amd import _ffi, debug
    assert sorted(_ffi.SYMBOLS) == _declared_functions()
    assert sorted(debug.DEBUG_SYMBOLS) == _declared_functions(DEBUG_HEADER)
    # the product header declares no test hook, and the two tables do not overlap
    assert not [n for n in _ffi.SYMBOLS if n.startswith("t2d_debug_")] and not set(_ffi.SYMBOLS) & set(debug.DEBUG_SYMBOLS)
    assert all(n.startswith("t2d_debug_") for n in debug.DEBUG_SYMBOLS)


def test_shared_library_exports_every_declared_symbol():
    from tactics2d_amd import build, _ffi
    build.build()
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} not exported by {_ffi.LIB_PATH}"
    lib.t2d_abi_version.restype = ctypes.c_int
    assert lib.t2d_abi_version() == _enum_values()["T2D_ABI_VERSION"]


def test_the_product_library_exports_no_test_hook_and_the_debug_library_exports_both_headers():
    """include/t2d_debug.h (fault injection, a gather delay, the stand-in policy and its closed-loop runner, placement maps)
    exists only in libt2d_hip_debug.so; libt2d_hip.so -- what the reference-side binding of INTEGRATION.md loads -- exports
    exactly the product ABI."""
    from tactics2d_amd import build, _ffi, debug
    build.build()
    build.build_debug_lib()
    prod = {n for n in _exported(_ffi.LIB_PATH) if n.startswith("t2d_")}
    assert not [n for n in prod if "debug" in n], sorted(n for n in prod if "debug" in n)
    assert prod == set(_declared_functions()), sorted(prod ^ set(_declared_functions()))
    dbg = {n for n in _exported(debug.DEBUG_LIB_PATH) if n.startswith("t2d_")}
    assert dbg == set(_declared_functions()) | set(_declared_functions(DEBUG_HEADER))


def test_no_product_module_imports_the_debug_module():
    pkg = os.path.join(ROOT, "tactics2d_amd")
    for f in sorted(os.listdir(pkg)):
        if f.endswith(".py") and f != "debug.py":
            src = open(os.path.join(pkg, f)).read()
            assert not re.search(r"^\s*(from\s+\.\s+import\s+.*\bdebug\b|from\s+\.debug\b|import\s+tactics2d_amd\.debug|from\s+tactics2d_amd\s+import\s+.*\bdebug\b|from\s+tactics2d_amd\.debug)", src, re.M), f
            assert "t2d_debug_" not in src.replace("t2d_debug.h", ""), f"{f} names a test hook"


def test_status_config_struct_matches_oracle_mirror():
    from oracle.oracle import StatusConfig as O
    from tactics2d_amd._ffi import StatusConfig as P
    assert [(n, t) for n, t in O._fields_] == [(n, t) for n, t in P._fields_]
    assert ctypes.sizeof(P) == 64


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a HIP device the product must raise -- there is no CPU compute path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tactics2d_amd import _ffi
    from tactics2d_amd.pool import ParticipantPool
    with pytest.raises(_ffi.T2DError) as ei:
        ParticipantPool(4, 2)
    assert ei.value.code == _ffi.ERR_HIP and "hip" in str(ei.value).lower()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tactics2d_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "t2d_oracle" not in src.replace("oracle/t2d_oracle.c", "").replace("oracle t2do_", ""), f


def test_only_tests_smoke_and_the_cpu_baseline_touch_the_oracle():
    """oracle/ is test infrastructure: besides tests/, only __graft_entry__ (build + smoke) and bench.py's cpu_baseline
    leg may import it; the helper scripts under scripts/ must not."""
    for f in sorted(os.listdir(os.path.join(ROOT, "scripts"))):
        if f.endswith((".py", ".sh")):
            src = open(os.path.join(ROOT, "scripts", f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
            assert "libt2d_oracle" not in src and "t2d_oracle" not in src, f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"(from|import)\s+oracle\b", bench)]
    assert uses, "bench.py's cpu_baseline leg is expected to time the oracle"
    lo = bench.index("def _cpu_leg")          # the two functions of the cpu_baseline leg
    hi = bench.index("\ndef ", bench.index("def cpu_baseline") + 1)
    assert all(lo < u < hi for u in uses), "the oracle may only be imported inside the cpu_baseline leg"


def test_step_form_names_follow_the_header():
    """pool.step_form() turns t2d_step_form's T2D_FORM_* value into a name: one name per enumerator, in the header's order"""
    import re
    from tactics2d_amd.pool import ParticipantPool
    vals = _enum_values()
    forms = sorted((v, k) for k, v in vals.items() if k.startswith("T2D_FORM_"))
    assert [v for v, _ in forms] == list(range(len(forms)))
    assert len(ParticipantPool.STEP_FORMS) == len(forms)
    for (v, k), name in zip(forms, ParticipantPool.STEP_FORMS):
        want = k[len("T2D_FORM_"):].lower()
        assert name == want or name == {"step": "step", "unfused": "unfused"}.get(want), (k, name)


def test_the_build_tracks_every_kernel_source_and_header():
    """tactics2d_amd.build.SOURCES / HEADERS decide when the library is rebuilt and what source_hash() -- the key the committed
    counter passes (profiles/traffic_latest.json) are valid for -- covers: every .hip / .h under csrc/ must be listed."""
    import os
    from tactics2d_amd import build as B
    on_disk = set(os.listdir(B.CSRC))
    assert {f for f in on_disk if f.endswith(".hip")} == set(B.SOURCES) | set(B.DEBUG_SOURCES)
    assert {f for f in on_disk if f.endswith(".h")} == {h for h in B.HEADERS if os.sep not in h and "/" not in h}
