"""ctypes mirror of include/qmgpu.h (the C ABI of the HIP library).

Plumbing only: structures, constants and the library loader.  The library is built in-tree by
``__graft_entry__.build()`` (hipcc, gfx950) as ``qm_door_amd/libqmgpu.so``; there is no CPU fallback --
if the shared object is missing, importing the compute entry points raises.
"""
import ctypes as C
import os

NX, NU, NV, NJ, NB, NC = 30, 30, 24, 18, 19, 4
NTARGET, NRBD, NWBC_DEC, NWBC_OUT = 37, 55, 36, 54
MAX_EVENTS = 40
NSTATS = 10
WBC_STATE_WORDS = 48   # uint64 words per instance of WbcArgs.working_set

OK = 0
ERR_INVALID_ARGUMENT, ERR_FILE_NOT_FOUND, ERR_PARSE, ERR_UNSUPPORTED_MODEL = 1, 2, 3, 4
ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_NUMERICAL = 5, 6, 7, 8

MODE_NAMES = {"FLY": 0, "RH": 1, "LH": 2, "LH_RH": 3, "RF": 4, "RF_RH": 5, "RF_LH": 6, "RF_LH_RH": 7, "LF": 8, "LF_RH": 9,
              "LF_LH": 10, "LF_LH_RH": 11, "LF_RF": 12, "LF_RF_RH": 13, "LF_RF_LH": 14, "STANCE": 15}

d, i32 = C.c_double, C.c_int32


class Model(C.Structure):
    _fields_ = [
        ("parent", i32 * NB), ("axis", i32 * NB), ("joint_offset", (d * 3) * NB), ("mass", d * NB), ("com", (d * 3) * NB),
        ("inertia", (d * 6) * NB), ("foot_body", i32 * NC), ("foot_offset", (d * 3) * NC), ("ee_body", i32), ("ee_offset", d * 3),
        ("q_lower", d * NJ), ("q_upper", d * NJ), ("effort_limit", d * NJ), ("velocity_limit", d * NJ), ("total_mass", d),
    ]


class Settings(C.Structure):
    _fields_ = [
        ("position_error_gain", d), ("phase_transition_stance_time", d),
        ("liftoff_velocity", d), ("touchdown_velocity", d), ("swing_height", d), ("touchdown_after_horizon", d), ("swing_time_scale", d),
        ("dt", d), ("time_horizon", d), ("delta_tol", d), ("g_max", d), ("g_min", d), ("alpha_decay", d), ("alpha_min", d), ("gamma_c", d),
        ("armijo_factor", d), ("cost_tol", d), ("sqp_iterations", i32), ("reserved0", i32),
        ("initial_state", d * NX), ("Q", d * (NX * NX)), ("R_task", d * (NU * NU)),
        ("ee_mu_position", d), ("ee_mu_orientation", d), ("ee_final_mu_position", d), ("ee_final_mu_orientation", d),
        ("friction_coefficient", d), ("friction_barrier_mu", d), ("friction_barrier_delta", d), ("friction_regularization", d),
        ("friction_hessian_shift", d), ("joint_pos_barrier_mu", d), ("joint_pos_barrier_delta", d), ("joint_vel_barrier_mu", d),
        ("joint_vel_barrier_delta", d), ("arm_vel_lower", d * 6), ("arm_vel_upper", d * 6),
        ("com_height", d), ("default_joint_state", d * NJ), ("target_displacement_velocity", d), ("target_rotation_velocity", d),
        ("wbc_friction_coefficient", d), ("kp_swing", d), ("kd_swing", d), ("kp_base_height", d), ("kd_base_height", d),
        ("kp_base_linear", d), ("kd_base_linear", d), ("kp_base_angular", d), ("kd_base_angular", d),
        ("kp_arm_joint", d * 6), ("kd_arm_joint", d * 6), ("kp_ee_linear", d * 3), ("kd_ee_linear", d * 3),
        ("kp_ee_angular", d * 3), ("kd_ee_angular", d * 3), ("gravity", d), ("ee_contact_stiffness", d), ("ee_force_mu", d), ("ddp_min_step", d), ("ddp_max_step", d), ("ddp_constraint_penalty", d),
    ]


class Problem(C.Structure):
    _fields_ = [("model", Model), ("settings", Settings)]


class Gait(C.Structure):
    _fields_ = [("num_modes", i32), ("modes", i32 * MAX_EVENTS), ("switching_times", d * (MAX_EVENTS + 1))]


class MpcArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("num_nodes", i32), ("num_target_knots", i32), ("line_search", i32),
        ("t0", C.c_void_p), ("x0", C.c_void_p), ("time_grid", C.c_void_p), ("target_times", C.c_void_p), ("target_states", C.c_void_p),
        ("sched_num_events", C.c_void_p), ("sched_event_times", C.c_void_p), ("sched_modes", C.c_void_p),
        ("warm_x", C.c_void_p), ("warm_u", C.c_void_p),
        ("out_t", C.c_void_p), ("out_x", C.c_void_p), ("out_u", C.c_void_p), ("out_mode", C.c_void_p), ("out_stats", C.c_void_p), ("ee_contact_ref", C.c_void_p), ("algorithm", i32), ("reserved1", i32),
    ]


class WbcArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("variant", i32),
        ("state_desired", C.c_void_p), ("input_desired", C.c_void_p), ("rbd_measured", C.c_void_p), ("mode", C.c_void_p),
        ("period", C.c_void_p), ("time", C.c_void_p), ("input_last", C.c_void_p), ("out", C.c_void_p), ("out_status", C.c_void_p), ("ee_force", C.c_void_p),
        ("working_set", C.c_void_p),
    ]


class FrontendArgs(C.Structure):
    _fields_ = [
        ("batch", i32),
        ("rbd_measured", C.c_void_p), ("time", C.c_void_p), ("yaw_last", C.c_void_p), ("command_kind", C.c_void_p), ("command", C.c_void_p),
        ("last_ee_target", C.c_void_p), ("feet_height", C.c_void_p),
        ("arm_dist", d), ("start_x", d), ("start_y", d), ("start_psi", d),
        ("x0", C.c_void_p), ("target_times", C.c_void_p), ("target_states", C.c_void_p),
    ]


PKG_DIR = os.path.dirname(os.path.abspath(__file__))
DATA_DIR = os.path.join(PKG_DIR, "data")
LIB_PATH = os.path.join(PKG_DIR, "libqmgpu.so")

# every symbol include/qmgpu.h declares
F64, F32 = 0, 1   # qmgpu_dtype

SYMBOLS = [
    "qmgpu_strerror", "qmgpu_last_error", "qmgpu_load_problem", "qmgpu_load_gait", "qmgpu_mode_from_string", "qmgpu_tile_gait", "qmgpu_switch_gait", "qmgpu_time_grid_with_events", "qmgpu_warm_start_batch",
    "qmgpu_create", "qmgpu_create_ex", "qmgpu_destroy", "qmgpu_set_stream", "qmgpu_synchronize", "qmgpu_get_input_weight", "qmgpu_mpc_solve_batch",
    "qmgpu_policy_eval_batch", "qmgpu_frontend_batch", "qmgpu_wbc_solve_batch", "qmgpu_cycle_batch", "qmgpu_debug_get_lq", "qmgpu_last_kernel_ms",
    "qmgpu_set_overlap", "qmgpu_join_wbc", "qmgpu_enable_timing", "qmgpu_enable_debug", "qmgpu_debug_poison", "qmgpu_kernel_ms_mean", "qmgpu_kernel_ms_history", "qmgpu_pack_results", "qmgpu_update_settings", "qmgpu_gait_schedule_batch",
]

_lib = None


def _preload_hip_runtime():
    """Bind to the HIP runtime PyTorch-ROCm ships (if present) before libqmgpu.so is mapped.

    torch/lib/libamdhip64.so carries SONAME libamdhip64.so.7 but no file of that name, so the dynamic loader would
    otherwise satisfy libqmgpu.so's NEEDED entry from /opt/rocm and the process would end up with two HIP/HSA runtimes,
    of which only the first one initialised sees the GPU.  Loading torch's copy first makes both users share it.
    """
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def load_library(path=None):
    """Load the HIP library. Fails loudly when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                           "qm_door_amd has no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(p)
    lib.qmgpu_strerror.restype = C.c_char_p
    lib.qmgpu_strerror.argtypes = [C.c_int]
    lib.qmgpu_last_error.restype = C.c_char_p
    lib.qmgpu_load_problem.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(Problem)]
    lib.qmgpu_load_gait.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Gait)]
    lib.qmgpu_mode_from_string.argtypes = [C.c_char_p]
    lib.qmgpu_frontend_batch.argtypes = [C.c_void_p, C.POINTER(FrontendArgs)]
    lib.qmgpu_tile_gait.argtypes = [C.POINTER(Gait), d, d, d, C.POINTER(i32), C.POINTER(d), C.POINTER(i32)]
    lib.qmgpu_time_grid_with_events.argtypes = [d, d, d, i32, C.POINTER(d), i32, C.POINTER(i32), C.POINTER(d)]
    lib.qmgpu_create.argtypes = [C.POINTER(Problem), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.qmgpu_destroy.argtypes = [C.c_void_p]
    lib.qmgpu_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.qmgpu_synchronize.argtypes = [C.c_void_p]
    lib.qmgpu_set_overlap.argtypes = [C.c_void_p, C.c_int]
    lib.qmgpu_join_wbc.argtypes = [C.c_void_p]
    lib.qmgpu_get_input_weight.argtypes = [C.c_void_p, C.POINTER(d)]
    lib.qmgpu_mpc_solve_batch.argtypes = [C.c_void_p, C.POINTER(MpcArgs)]
    lib.qmgpu_policy_eval_batch.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8
    lib.qmgpu_warm_start_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.qmgpu_wbc_solve_batch.argtypes = [C.c_void_p, C.POINTER(WbcArgs)]
    lib.qmgpu_cycle_batch.argtypes = [C.c_void_p, C.POINTER(MpcArgs), C.c_void_p, C.POINTER(WbcArgs)]
    lib.qmgpu_debug_get_lq.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 10 + [C.POINTER(i32)]
    lib.qmgpu_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(d)]
    lib.qmgpu_enable_timing.argtypes = [C.c_void_p, C.c_int]
    lib.qmgpu_kernel_ms_mean.argtypes = [C.c_void_p, C.c_int, C.POINTER(d)]
    lib.qmgpu_kernel_ms_history.argtypes = [C.c_void_p, C.c_int, C.POINTER(d)]
    lib.qmgpu_pack_results.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
    lib.qmgpu_enable_debug.argtypes = [C.c_void_p, C.c_int]
    lib.qmgpu_debug_poison.argtypes = [C.c_void_p]
    if path is None:
        _lib = lib
    return lib


class QmGpuError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__(f"qmgpu status {status}: {detail}")
        self.status = status


def check(lib, status):
    if status != OK:
        raise QmGpuError(status, (lib.qmgpu_strerror(status) or b"").decode() + " -- " + (lib.qmgpu_last_error() or b"").decode())
