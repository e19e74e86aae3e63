"""Participant parameter templates -> kernel parameter rows.

Numeric tables restate tactics2d/participant/element/participant_template.py:42-257 (vehicle,
cyclist, pedestrian dimensions and limits) and the way each element class turns a template into
a physics model:
    Vehicle.load_from_template   vehicle.py:179-221  (max_accel = round(100/3.6/t_0_100, 3),
                                 speed_range = (-16.67, max_speed), accel_range = (-max_decel, max_accel),
                                 max_steer = round(pi/6, 3) = 0.524, vehicle.py:111)
    lf = L/2 - front_overhang, lr = L/2 - rear_overhang      envs/parking.py:321-322
    Cyclist                      cyclist.py:85-97  (kinematics, lf = lr = L/2, speed (0, vmax))
    Pedestrian                   pedestrian.py:74-92 (PointMass, ranges (-v, v) -> [0, v]; radius = width/2)
"""
import numpy as np

from . import layout as L
from .physics import PointMass, SingleTrackDynamics, SingleTrackKinematics

# name: (length, width, height, wheel_base, front_overhang, rear_overhang, kerb_weight,
#        max_speed, t_0_100, max_decel)
VEHICLE_TEMPLATE = {
    "mini_car": (3.540, 1.641, 1.489, 2.420, 0.585, 0.535, 1070, 44.44, 14.4, 10.0),
    "small_car": (4.053, 1.751, 1.461, 2.548, 0.824, 0.681, 1565, 52.78, 11.2, 10.0),
    "medium_car": (4.284, 1.799, 1.452, 2.637, 0.880, 0.767, 1620, 69.44, 8.9, 11.0),
    "large_car": (4.866, 1.832, 1.477, 2.871, 0.955, 1.040, 1735, 58.33, 8.4, 11.0),
    "executive_car": (5.050, 1.886, 1.475, 3.024, 0.921, 1.105, 2175, 63.89, 8.1, 11.3),
    "luxury_car": (5.302, 1.945, 1.488, 3.128, 0.989, 1.185, 2520, 69.44, 6.7, 11.3),
    "sports_coupe": (4.788, 1.916, 1.381, 2.720, 0.830, 1.238, 1740, 63.89, 5.3, 10.4),
    "multi_purpose_car": (5.155, 1.995, 1.740, 3.090, 0.935, 1.130, 2095, 66.67, 9.4, 10.3),
    "sports_utility_car": (4.828, 1.943, 1.792, 2.915, 0.959, 0.954, 2200, 88.89, 3.8, 10.29),
}
# name: (length, width, height, max_steer, max_speed, max_accel, max_decel)
CYCLIST_TEMPLATE = {
    "cyclist": (1.80, 0.65, 1.70, 1.05, 22.78, 5.8, 7.8),
    "moped": (2.00, 0.70, 1.70, 0.35, 13.89, 3.5, 7.0),
    "motorcycle": (2.40, 0.80, 1.70, 0.44, 75.00, 5.0, 10.0),
}
# name: (length, width, height, max_speed, max_accel)
PEDESTRIAN_TEMPLATE = {
    "adult_male": (0.24, 0.40, 1.75, 7.0, 1.5),
    "adult_female": (0.22, 0.37, 1.65, 6.0, 1.5),
    "children_six_year_old": (0.18, 0.25, 1.16, 3.5, 1.0),
    "children_ten_year_old": (0.20, 0.35, 1.42, 4.5, 1.0),
}
MAX_STEER = float(np.round(np.pi / 6, 3))  # vehicle.py:111


def vehicle_model(name, model="kinematics", interval=100, delta_t=None, speed_range=None,
                  accel_range=None, steer_range=None):
    """The physics model a reference `Vehicle` of this template would be driven by."""
    Ln, W, Hh, wb, fo, ro, kerb, vmax, t0100, decel = VEHICLE_TEMPLATE[name]
    max_accel = float(np.round(100 * 1000 / 3600 / t0100, 3))  # vehicle.py:199-201
    lf, lr = Ln / 2 - fo, Ln / 2 - ro
    sr = (-MAX_STEER, MAX_STEER) if steer_range is None else steer_range
    vr = (-16.67, vmax) if speed_range is None else speed_range
    ar = (-decel, max_accel) if accel_range is None else accel_range
    if model == "kinematics":
        return SingleTrackKinematics(lf, lr, sr, vr, ar, interval, delta_t)
    if model == "dynamics":  # mass = kerb weight, cg height = height / 2 (docstring advice, :83-84)
        return SingleTrackDynamics(lf, lr, float(kerb), Hh / 2, steer_range=sr, speed_range=vr,
                                   accel_range=ar, interval=interval, delta_t=delta_t)
    raise ValueError(model)


def vehicle_row(name, model="kinematics", **kw):
    Ln, W = VEHICLE_TEMPLATE[name][:2]
    return vehicle_model(name, model, **kw).param_row(L.SHAPE_OBB, Ln, W)


def cyclist_row(name, interval=100, delta_t=None):
    Ln, W, _, steer, vmax, amax, decel = CYCLIST_TEMPLATE[name]
    m = SingleTrackKinematics(Ln / 2, Ln / 2, (-steer, steer), (0, vmax), (-decel, amax), interval, delta_t)
    return m.param_row(L.SHAPE_OBB, Ln, W)


def pedestrian_row(name, interval=100, delta_t=None):
    Ln, W, _, vmax, amax = PEDESTRIAN_TEMPLATE[name]
    m = PointMass((-vmax, vmax), (-amax, amax), interval, delta_t)
    return m.param_row(L.SHAPE_CIRCLE, Ln, W)


def full_type_table(interval=100):
    """All 25 reference participant types as one table (<= 32 rows): 9 kinematic vehicles,
    9 dynamic vehicles, 3 cyclists, 4 pedestrians.  Returns (rows, names)."""
    rows, names = [], []
    for n in VEHICLE_TEMPLATE:
        rows.append(vehicle_row(n, "kinematics", interval=interval)); names.append(n + ":kin")
    for n in VEHICLE_TEMPLATE:
        rows.append(vehicle_row(n, "dynamics", interval=interval)); names.append(n + ":dyn")
    for n in CYCLIST_TEMPLATE:
        rows.append(cyclist_row(n, interval)); names.append(n)
    for n in PEDESTRIAN_TEMPLATE:
        rows.append(pedestrian_row(n, interval)); names.append(n)
    return np.array(rows), names
