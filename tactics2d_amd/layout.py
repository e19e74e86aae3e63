"""Numeric constants of the C ABI (include/t2d.h), mirrored for the Python host side.
tests/test_layout.py parses the header and checks that the two agree."""
ABI_VERSION = 12   # T2D_ABI_VERSION: _ffi.lib() refuses a libt2d_hip.so built from another header
# parameter-row columns
P_MODEL, P_LF, P_LR, P_WB = 0, 1, 2, 3
P_STEER_LO, P_STEER_HI, P_SPEED_LO, P_SPEED_HI, P_ACCEL_LO, P_ACCEL_HI = 4, 5, 6, 7, 8, 9
P_RANGE_FLAGS, P_MASS, P_MASS_HEIGHT, P_MU, P_IZ, P_CF, P_CR = 10, 11, 12, 13, 14, 15, 16
P_DELTA_T_MS, P_SHAPE, P_LENGTH, P_WIDTH = 17, 18, 19, 20
PARAM_COLS = 24
MAX_TYPES = 32
RANGE_STEER, RANGE_SPEED, RANGE_ACCEL = 1, 2, 4
MODEL_KINEMATICS, MODEL_DYNAMICS, MODEL_POINTMASS, MODEL_DRIFT, MODEL_POINTMASS_EULER = 0, 1, 2, 3, 4
P_DRIFT_TSB, P_DRIFT_TSE, P_DRIFT_RADIUS, P_DRIFT_IYW = 15, 16, 22, 23   # SingleTrackDrift rows only
P_DT_S, P_SUBSTEPS = 22, 23   # rows of the other models: derived by the library (sub-step in s, sub-step counts of the launch)
MAX_INTERVAL_MS = 32767
SHAPE_OBB, SHAPE_CIRCLE = 0, 1
# fields
F_X, F_Y, F_HEADING, F_SPEED, F_VX, F_VY, F_ACT0, F_ACT1, F_IDS, F_FLAGS = range(10)
F_APPLIED0, F_APPLIED1, F_ENV_FLAGS, F_CNT_STEP, F_FRAME_MS, F_STATUS, F_REWARD = range(10, 17)
F_RECORD = 17
F_IOU = 18
F_CNT_NO_ACTION = 19
F_LIDAR = 20
F_LEADER = 21
F_OMEGA_F, F_OMEGA_R = 22, 23
F_COUNT = 24
FIELD_DTYPES = {
    F_X: "float32", F_Y: "float32", F_HEADING: "float32", F_SPEED: "float32", F_VX: "float32",
    F_VY: "float32", F_ACT0: "float32", F_ACT1: "float32", F_IDS: "uint32", F_FLAGS: "uint32",
    F_APPLIED0: "float32", F_APPLIED1: "float32", F_ENV_FLAGS: "uint32", F_CNT_STEP: "int32",
    F_FRAME_MS: "int32", F_STATUS: "uint8", F_REWARD: "float32", F_RECORD: "uint32", F_IOU: "float32", F_CNT_NO_ACTION: "int32", F_LIDAR: "float32",
    F_LEADER: "int32", F_OMEGA_F: "float32", F_OMEGA_R: "float32",
}
PER_ENV_FIELDS = (F_ENV_FLAGS, F_CNT_STEP, F_FRAME_MS, F_STATUS, F_REWARD, F_RECORD, F_IOU, F_CNT_NO_ACTION, F_LIDAR)
# event bits
FLAG_COLLISION_DYNAMIC, FLAG_COLLISION_STATIC, FLAG_OUT_BOUND, FLAG_OFF_LANE = 1, 2, 4, 8
RECORD_RING = 64
OUT_VELOCITY, OUT_APPLIED, OUT_ALL = 1, 2, 3
SAFE_RECTS = 4
MAX_POLY_VERTS = 8
MAX_AGENTS = 256
# IDM controller parameter sets (t2d_set_idm)
IDM_DESIRED_SPEED, IDM_TIME_HEADWAY, IDM_MIN_SPACING, IDM_MAX_ACCEL, IDM_COMF_DECEL, IDM_DELTA = range(6)
IDM_LANE_HALF_WIDTH, IDM_HORIZON = 6, 7
IDM_COLS = 8
IDM_NONE = 255
IDM_LEADER_FREE, IDM_LEADER_SEARCH = -1, -2
# host-frame sections (t2d_frame_config)
FRAME_LIDAR, FRAME_TARGET, FRAME_ZEROCOPY = 1, 2, 4
