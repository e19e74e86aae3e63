#!/usr/bin/env python3
"""Golden data for the participant parameter tables (row a9) and Map.boundary (row a15), from the reference itself.

TEST INFRASTRUCTURE (build container only; the reference tree is mounted read-only).
  * `tactics2d/participant/element/participant_template.py` is a pure-data module (three dicts; it imports only `tabulate`), but
    its package imports shapely: the FILE is loaded with importlib, nothing else of the package.  The dicts' numeric entries go
    to tests/golden/templates.json as they are.
  * `Vehicle.load_from_template` and `Vehicle.get_pose` (`participant/element/vehicle.py:179-221, 263-281`) are taken from the parsed
    file and EXECUTED on a blank attribute holder, with recorders where they hand their results to shapely (`LinearRing`: the
    bounding box and its vertex order; `affine_transform`: the matrix) -> tests/golden/vehicle_templates_loaded.json: max_accel
    (the rounded 0-100 rule), speed / accel ranges, bbox, and six poses per template.
  * the literal constants of `tactics2d/envs/parking.py` (module level, `ParkingEnv`, `_ParkingScenarioManager`, the defaults of
    `ParkingEnv.__init__`: read from the parsed file -- the module needs gymnasium) and the enums of `tactics2d/traffic/status.py`
    (loaded as a file) -> tests/golden/parking_constants.json.
  * `Map.boundary` (`tactics2d/map/element/map.py:92-167`) reads `.nodes` / `.lanes` / `.areas` / `.roadlines` of the map and,
    of every element, `.x` / `.y` or `.geometry(.exterior).coords`: the property's getter is taken from the parsed file and
    EXECUTED, unmodified, on maps made of plain data holders with seeded coordinates (the module itself cannot be imported:
    shapely).  Expected boundaries go to tests/golden/map_boundary.json beside the coordinates.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_tables.py [--ref /root/reference]
"""
import argparse
import ast
import importlib.util
import json
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    # ---- templates -------------------------------------------------------------------------------------------------------------
    path = os.path.join(args.ref, "tactics2d", "participant", "element", "participant_template.py")
    spec = importlib.util.spec_from_file_location("t2d_ref_participant_template", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    numeric = lambda d: {k: {f: v for f, v in row.items() if isinstance(v, (int, float))} for k, row in d.items()}
    templates = dict(vehicle=numeric(m.VEHICLE_TEMPLATE), cyclist=numeric(m.CYCLIST_TEMPLATE), pedestrian=numeric(m.PEDESTRIAN_TEMPLATE))
    os.makedirs(OUT, exist_ok=True)
    json.dump(templates, open(os.path.join(OUT, "templates.json"), "w"), indent=1, sort_keys=True)
    print("templates:", {k: len(v) for k, v in templates.items()})
    # ---- Map.boundary ------------------------------------------------------------------------------------------------------------
    mp = os.path.join(args.ref, "tactics2d", "map", "element", "map.py")
    tree = ast.parse(open(mp).read(), filename=mp)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Map")
    getter = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "boundary"
                  and any(isinstance(d, ast.Name) and d.id == "property" for d in n.decorator_list))
    getter.decorator_list = []      # (called as a plain function of the map below)
    ns = {"np": np}
    exec(compile(ast.Module(body=[getter], type_ignores=[]), mp, "exec"), ns)
    boundary_of = ns["boundary"]
    print(f"executing Map.boundary: lines {getter.lineno}-{getter.end_lineno}")
    rng = np.random.default_rng(20261003)
    cases = []
    for k in range(40):
        scale = float(rng.choice([10.0, 300.0, 5000.0]))
        off = rng.uniform(-scale, scale, 2)
        poly = lambda n: (off + rng.uniform(-scale, scale, (n, 2))).round(int(rng.integers(0, 6)))
        nodes = [poly(1)[0].tolist() for _ in range(int(rng.integers(0, 5)))]
        lanes = [poly(int(rng.integers(2, 9))).tolist() for _ in range(int(rng.integers(0, 4)))]
        areas = [poly(int(rng.integers(3, 7))).tolist() for _ in range(int(rng.integers(0, 4)))]
        roadlines = [poly(int(rng.integers(2, 6))).tolist() for _ in range(int(rng.integers(0, 3)))]
        if k == 0:
            nodes, lanes, areas, roadlines = [], [], [], []                      # the empty map: (0, 0, 0, 0)
        if k == 1:
            nodes, lanes, areas, roadlines = [[2.0, -3.0]], [], [], []           # integers stay what they are
        holder = lambda pts, ring=False: types.SimpleNamespace(geometry=(types.SimpleNamespace(exterior=types.SimpleNamespace(coords=pts))
                                                                         if ring else types.SimpleNamespace(coords=pts)))
        map_ = types.SimpleNamespace(
            _boundary=None, _min_x=None, _max_x=None, _min_y=None, _max_y=None,
            nodes={i: types.SimpleNamespace(x=p[0], y=p[1]) for i, p in enumerate(nodes)},
            lanes={i: holder(p) for i, p in enumerate(lanes)}, areas={i: holder(p, True) for i, p in enumerate(areas)},
            roadlines={i: holder(p) for i, p in enumerate(roadlines)})
        b = boundary_of(map_)
        cases.append(dict(nodes=nodes, lanes=lanes, areas=areas, roadlines=roadlines, boundary=[float(v) for v in b]))
    json.dump(cases, open(os.path.join(OUT, "map_boundary.json"), "w"))
    print(f"{len(cases)} maps -> tests/golden/map_boundary.json")
    # ---- Vehicle.load_from_template / get_pose (participant/element/vehicle.py:179-221, 263-281) -----------------------------------
    # executed as they stand on a blank attribute holder; LinearRing records the vertex list it is given (the bounding box and
    # its vertex ORDER), affine_transform records the matrix and applies shapely's documented rule for a 6-element matrix
    # [a, b, d, e, xoff, yoff]: x' = a x + b y + xoff, y' = d x + e y + yoff
    vp = os.path.join(args.ref, "tactics2d", "participant", "element", "vehicle.py")
    vtree = ast.parse(open(vp).read(), filename=vp)
    vcls = next(n for n in vtree.body if isinstance(n, ast.ClassDef) and n.name == "Vehicle")
    import logging
    seen = types.SimpleNamespace(matrix=None)

    def affine_transform(ring, mat):
        seen.matrix = [float(v) for v in mat]
        a, b, d, e, xo, yo = mat
        return [[a * x + b * y + xo, d * x + e * y + yo] for x, y in ring]
    vns = {"np": np, "logging": logging, "LinearRing": lambda pts: [list(map(float, p)) for p in pts], "affine_transform": affine_transform,
           "VEHICLE_TEMPLATE": m.VEHICLE_TEMPLATE, "EURO_SEGMENT_MAPPING": m.EURO_SEGMENT_MAPPING, "EPA_MAPPING": m.EPA_MAPPING,
           "NCAP_MAPPING": m.NCAP_MAPPING, "State": object, "Tuple": tuple}
    for name in ("load_from_template", "get_pose"):
        fn = next(n for n in vcls.body if isinstance(n, ast.FunctionDef) and n.name == name)
        fn.returns = None
        for a in fn.args.args:
            a.annotation = None       # (annotations name shapely / typing classes: dropped, the bodies are untouched)
        exec(compile(ast.Module(body=[fn], type_ignores=[]), vp, "exec"), vns)
        print(f"executing Vehicle.{name}: lines {fn.lineno}-{fn.end_lineno}")
    fields = ("length", "width", "height", "kerb_weight", "wheel_base", "front_overhang", "rear_overhang", "max_speed", "max_accel",
              "max_decel", "max_steer", "driven_mode")
    vehicles = {}
    for name in m.VEHICLE_TEMPLATE:
        holder = types.SimpleNamespace(**{k: None for k in fields}, _bbox=None, speed_range=None, accel_range=None)
        try:
            vns["load_from_template"](holder, name)
        except TypeError as exc:
            # a quirk of the reference, recorded as it is: "multi_purpose_car" is also a key of EPA_MAPPING (-> "minivan", which no
            # template holds), so loading that template BY NAME falls through to the defaults and fails on -None
            vehicles[name] = dict(error=f"{type(exc).__name__}: {exc}")
            continue
        poses = []
        for _ in range(6):
            x, y, h = (float(np.float32(v)) for v in (rng.uniform(-200, 200), rng.uniform(-200, 200), rng.uniform(-7, 7)))
            holder.trajectory = types.SimpleNamespace(get_state=lambda frame, x=x, y=y, h=h: types.SimpleNamespace(heading=h, location=(x, y)))
            pose = vns["get_pose"](holder, 0)
            poses.append(dict(x=x, y=y, heading=h, matrix=seen.matrix, pose=pose))
        vehicles[name] = dict(max_accel=float(holder.max_accel), speed_range=[float(v) for v in holder.speed_range],
                              accel_range=[float(v) for v in holder.accel_range], bbox=holder._bbox, poses=poses)
    json.dump(vehicles, open(os.path.join(OUT, "vehicle_templates_loaded.json"), "w"))
    print(f"{len(vehicles)} vehicles x 6 poses -> tests/golden/vehicle_templates_loaded.json")
    # ---- the literal constants of envs/parking.py and traffic/status.py (the module cannot be imported: gymnasium, shapely) -------
    ep = os.path.join(args.ref, "tactics2d", "envs", "parking.py")
    etree = ast.parse(open(ep).read(), filename=ep)
    consts = {}

    def literal_assigns(body, prefix, env):
        for st in body:
            if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name):
                try:
                    val = eval(compile(ast.Expression(st.value), ep, "eval"), {"__builtins__": {"int": int}}, dict(env))
                except Exception:      # noqa: BLE001 -- not a literal (a call into gymnasium ...): not a constant
                    continue
                env[st.targets[0].id] = val
                consts[prefix + st.targets[0].id] = val
    module_env = {}
    literal_assigns(etree.body, "", module_env)
    pcls = next(n for n in etree.body if isinstance(n, ast.ClassDef) and n.name == "ParkingEnv")
    literal_assigns(pcls.body, "ParkingEnv.", dict(module_env))
    mcls = next(n for n in pcls.body if isinstance(n, ast.ClassDef) and n.name == "_ParkingScenarioManager")
    literal_assigns(mcls.body, "ParkingEnv._ParkingScenarioManager.", dict(module_env))
    init = next(n for n in pcls.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
    defaults = dict(zip([a.arg for a in init.args.args][-len(init.args.defaults):],
                        [eval(compile(ast.Expression(dv), ep, "eval"), {"__builtins__": {"int": int}}) for dv in init.args.defaults]))
    consts["ParkingEnv.__init__.defaults"] = defaults
    sp = os.path.join(args.ref, "tactics2d", "traffic", "status.py")
    sspec = importlib.util.spec_from_file_location("t2d_ref_status", sp)
    sm = importlib.util.module_from_spec(sspec)
    sspec.loader.exec_module(sm)
    consts["ScenarioStatus"] = {e.name: int(e) for e in sm.ScenarioStatus}
    consts["TrafficStatus"] = {e.name: int(e) for e in sm.TrafficStatus}
    jsonable = lambda v: {str(k): jsonable(x) for k, x in v.items()} if isinstance(v, dict) else ([jsonable(x) for x in v] if isinstance(v, (list, tuple)) else v)
    json.dump(jsonable(consts), open(os.path.join(OUT, "parking_constants.json"), "w"), indent=1, sort_keys=True)
    print("constants:", sorted(consts))


if __name__ == "__main__":
    main()
