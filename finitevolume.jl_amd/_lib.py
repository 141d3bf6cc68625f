"""ctypes binding of libfvhip.so (include/fvhip.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible
the import / context creation fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(_HERE, "libfvhip.so")

FV_OK = 0
FV_ERR_ARG = 1
FV_ERR_SOURCE_AT_DIRICHLET = 2
FV_ERR_INDEX = 3
FV_ERR_NOMEM = 4
FV_ERR_HIP = 5
FV_ERR_STATE = 6
FV_ERR_DT = 7
FV_ERR_TOO_LARGE = 8
FV_ERR_COMM = 9

FV_STEP_FORWARD = 0
FV_STEP_ADJOINT = 1
FV_COMM_ID_BYTES = 128


class FVError(RuntimeError):
    """Julia's `error(msg)` / BoundsError at the reference's call sites."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class SolveInfo(C.Structure):
    _fields_ = [
        ("converged", C.c_int32),
        ("iters", C.c_int32),
        ("relres", C.c_double),
        ("bnorm", C.c_double),
        ("solve_ms", C.c_double),
        ("resnorm_len", C.c_int64),
    ]


c_ctx = C.c_void_p
c_prob = C.c_void_p
P = C.POINTER
_i64p = C.c_void_p  # arrays are passed as raw addresses (host or device)
_f64p = C.c_void_p

# name -> (restype, argtypes): every symbol include/fvhip.h declares
ABI_VERSION = 4  # FVHIP_ABI_VERSION of include/fvhip.h this binding was written against
FV_OPT_REORDER = 1
FV_OPT_LEAN_SETUP = 2
# the experimenter's panel (finitevolume.jl_amd/csrc/fv_tune.h): exported, but not part of include/fvhip.h
PRIVATE_SIGNATURES = {"fv_tune": (C.c_int, [C.c_int, C.c_int]),
                      "fv_comm_init_local": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int])}  # the loopback transport of the multi-rank rehearsals (tests)
SIGNATURES = {
    "fv_abi_version": (C.c_int, []),
    "fv_ctx_set_option": (C.c_int, [c_ctx, C.c_int, C.c_int]),
    "fv_ctx_get_option": (C.c_int, [c_ctx, C.c_int, P(C.c_int)]),
    "fv_ctx_create": (C.c_int, [C.c_int, P(c_ctx)]),
    "fv_ctx_destroy": (None, [c_ctx]),
    "fv_ctx_synchronize": (C.c_int, [c_ctx]),
    "fv_last_error": (C.c_char_p, [c_ctx]),
    "fv_device_info": (C.c_int, [c_ctx, C.c_char_p, C.c_int, P(C.c_int), P(C.c_int64)]),
    "fv_regulargrid_sizes": (C.c_int, [_i64p, P(C.c_int64), P(C.c_int64)]),
    "fv_regulargrid": (C.c_int, [c_ctx, _f64p, _f64p, _i64p, _f64p, _i64p, _i64p, _f64p, _f64p]),
    "fv_nodehycos2neighborhycos": (C.c_int, [c_ctx, C.c_int64, _i64p, _i64p, C.c_int64, _f64p, C.c_int, _f64p]),
    "fv_getfreenodes": (C.c_int, [c_ctx, C.c_int64, C.c_int64, _i64p, C.c_void_p, _i64p, P(C.c_int64)]),
    "fv_getnodei2dirichleti": (C.c_int, [c_ctx, C.c_int64, _f64p, C.c_int64, _i64p, _i64p, P(C.c_int64)]),
    "fv_problem_create": (C.c_int, [c_ctx, C.c_int64, C.c_int64, _i64p, _i64p, _f64p, C.c_int64, _i64p, P(c_prob)]),
    "fv_problem_create_regulargrid": (C.c_int, [c_ctx, _f64p, _f64p, _i64p, C.c_int64, _i64p, P(c_prob)]),
    "fv_problem_create_from_csc": (C.c_int, [c_ctx, C.c_int64, _i64p, _i64p, _f64p, P(c_prob)]),
    "fv_problem_destroy": (None, [c_prob]),
    "fv_problem_sizes": (C.c_int, [c_prob, P(C.c_int64), P(C.c_int64), P(C.c_int64), P(C.c_int64)]),
    "fv_problem_get_free_maps": (C.c_int, [c_prob, C.c_void_p, _i64p]),
    "fv_problem_get_grid": (C.c_int, [c_prob, _i64p, _i64p, _f64p, _f64p]),
    "fv_assemble": (C.c_int, [c_prob, C.c_int64, _f64p, _i64p, C.c_int, _f64p, _f64p, P(C.c_int64)]),
    "fv_get_csc": (C.c_int, [c_prob, _i64p, _i64p, _f64p]),
    "fv_get_b": (C.c_int, [c_prob, _f64p]),
    "fv_freenodes2nodes": (C.c_int, [c_prob, _f64p, _f64p]),
    "fv_solve_steady": (C.c_int, [c_prob, _f64p, C.c_double, C.c_int64, _f64p, _f64p, _f64p, C.c_int64, P(SolveInfo)]),
    "fv_transient_begin": (C.c_int, [c_prob, C.c_double, _f64p, _f64p]),
    "fv_state_alloc": (C.c_int, [c_prob, P(C.c_int32)]),
    "fv_state_free": (C.c_int, [c_prob, C.c_int32]),
    "fv_state_set_nodes": (C.c_int, [c_prob, C.c_int32, _f64p]),
    "fv_state_set_free": (C.c_int, [c_prob, C.c_int32, _f64p]),
    "fv_state_get_nodes": (C.c_int, [c_prob, C.c_int32, _f64p]),
    "fv_state_get_free": (C.c_int, [c_prob, C.c_int32, _f64p]),
    "fv_state_copy": (C.c_int, [c_prob, C.c_int32, C.c_int32]),
    "fv_state_norm2_diff": (C.c_int, [c_prob, C.c_int32, C.c_int32, P(C.c_double)]),
    "fv_transient_step": (C.c_int, [c_prob, C.c_int32, C.c_int32, C.c_double, _f64p, C.c_int, C.c_double, C.c_int64, P(SolveInfo)]),
    "fv_transient_run_fixed": (C.c_int, [c_prob, C.c_int32, C.c_double, C.c_int64, C.c_double, C.c_int64, C.c_void_p, P(SolveInfo), P(C.c_double)]),
    "fv_transient_run_adaptive": (C.c_int, [c_prob, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int64, _f64p, P(C.c_int64), P(C.c_int64), P(SolveInfo)]),
    "fv_spmv": (C.c_int, [c_prob, _f64p, C.c_double, _f64p]),
    "fv_bench_spmv": (C.c_int, [c_prob, C.c_double, C.c_int32, P(C.c_double)]),
    "fv_dot": (C.c_int, [c_prob, _f64p, _f64p, P(C.c_double)]),
    "fv_device_mem_info": (C.c_int, [c_ctx, P(C.c_int64), P(C.c_int64)]),
    "fv_precond_set": (C.c_int, [c_prob, C.c_int]),
    "fv_amg_configure": (C.c_int, [C.c_double, C.c_double, C.c_int, C.c_int]),
    "fv_amg_info": (C.c_int, [c_prob, P(C.c_int32), _i64p, _i64p, C.c_int32]),
    "fv_amg_apply": (C.c_int, [c_prob, _f64p, C.c_double, _f64p]),
    "fv_profile_enable": (C.c_int, [c_prob, C.c_int]),
    "fv_profile_get": (C.c_int, [c_prob, C.c_int, P(C.c_double), P(C.c_int64)]),
    "fv_spmv_form": (C.c_int, [c_prob, P(C.c_int32), P(C.c_int64)]),
    "fv_update_form": (C.c_int, [c_prob, P(C.c_int32)]),
    "fv_loop_form": (C.c_int, [c_prob, P(C.c_int32)]),
    "fv_step_form": (C.c_int, [c_prob, P(C.c_int32), P(C.c_int64), P(C.c_int64)]),
    "fv_fused_form": (C.c_int, [c_prob, P(C.c_int64), P(C.c_int32), P(C.c_int64)]),
    "fv_fused_traversal": (C.c_int, [c_prob, P(C.c_int32)]),
    "fv_problem_reorder_info": (C.c_int, [c_prob, P(C.c_int32), P(C.c_double), P(C.c_double), P(C.c_double)]),
    "fv_comm_unique_id": (C.c_int, [C.c_char_p]),
    "fv_comm_init": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_char_p]),
    "fv_comm_destroy": (C.c_int, [c_ctx]),
    "fv_comm_selftest": (C.c_int, [c_ctx, C.c_int64, P(C.c_int)]),
    "fv_comm_diag": (C.c_int, [c_ctx, C.c_int]),
    "fv_comm_diag_get": (C.c_int, [c_ctx, P(C.c_double), P(C.c_int64)]),
    "fv_dist_setup": (C.c_int, [c_prob, C.c_int, C.c_int, P(c_prob)]),
    "fv_param_gradient_integral": (C.c_int, [c_prob, C.c_int64, _f64p, _f64p, _f64p, C.c_int, C.c_int, _f64p, _f64p, _f64p]),
    "fv_trajectory_create": (C.c_int, [c_prob, P(C.c_void_p)]),
    "fv_trajectory_destroy": (C.c_int, [C.c_void_p]),
    "fv_trajectory_clear": (C.c_int, [C.c_void_p]),
    "fv_trajectory_push_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    "fv_trajectory_push_free": (C.c_int, [C.c_void_p, _f64p, C.c_double]),
    "fv_trajectory_size": (C.c_int, [C.c_void_p, P(C.c_int64)]),
    "fv_trajectory_times": (C.c_int, [C.c_void_p, _f64p, C.c_int64]),
    "fv_trajectory_get_free": (C.c_int, [C.c_void_p, C.c_int64, _f64p]),
    "fv_trajectory_get_nodes": (C.c_int, [C.c_void_p, C.c_int64, _f64p]),
    "fv_trajectory_eval_free": (C.c_int, [C.c_void_p, C.c_double, _f64p]),
    "fv_trajectory_reverse_time": (C.c_int, [C.c_void_p, C.c_double]),
    "fv_trajectory_record": (C.c_int, [c_prob, C.c_void_p, C.c_double]),
    "fv_observation_create": (C.c_int, [c_prob, C.c_int64, _i64p, C.c_int64, _f64p, _f64p, _f64p, P(C.c_void_p)]),
    "fv_observation_destroy": (C.c_int, [C.c_void_p]),
    "fv_observation_integral": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, P(C.c_double)]),
    "fv_adjoint_run": (C.c_int, [c_prob, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int64, C.c_int64,
                                 C.c_void_p, P(C.c_int64), P(C.c_int64), C.c_void_p]),
    "fv_param_gradient_integral_traj": (C.c_int, [c_prob, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, _f64p, C.c_int, _f64p, _f64p, _f64p]),
    "fv_param_jacobian_apply": (C.c_int, [c_prob, _f64p, _f64p, C.c_int, C.c_int, _f64p, _f64p, _f64p]),
    "fv_dist_setup_bounds": (C.c_int, [c_prob, C.c_int, C.c_int, _i64p, P(c_prob)]),
    "fv_problem_create_regulargrid_slab": (C.c_int, [c_ctx, _f64p, _f64p, _i64p, C.c_int64, _i64p, C.c_int64, C.c_int64, P(c_prob)]),
    "fv_problem_free_rows_before": (C.c_int, [c_prob, C.c_int64, P(C.c_int64)]),
    "fv_dist_plan_sizes": (C.c_int, [c_prob] + [P(C.c_int64)] * 7),
    "fv_dist_get_plan": (C.c_int, [c_prob] + [_i64p] * 7),
    "fv_dist_run_fixed": (C.c_int, [c_prob, C.c_double, C.c_int64, C.c_double, C.c_int64, C.c_void_p, P(SolveInfo), P(C.c_double)]),
    "fv_dist_solve_steady": (C.c_int, [c_prob, _f64p, C.c_double, C.c_int64, _f64p, P(SolveInfo)]),
    "fv_dist_step": (C.c_int, [c_prob, C.c_double, _f64p, C.c_double, C.c_int64, P(SolveInfo)]),
    "fv_dist_run_adaptive": (C.c_int, [c_prob, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int64, _f64p, P(C.c_int64), P(C.c_int64), P(SolveInfo)]),
    "fv_comm_stats": (C.c_int, [c_ctx, P(C.c_int64), P(C.c_int64), C.c_int]),
    "fv_dist_spmv": (C.c_int, [c_prob, _f64p, C.c_double, _f64p]),
    "fv_dist_spmv_halo": (C.c_int, [c_prob, _f64p, _f64p, C.c_double, _f64p]),
    "fv_dist_state_get": (C.c_int, [c_prob, _f64p]),
}

_lib = None


def legacy_tune(raw):
    """The library's panel has 18 keys since round 5 (csrc/fv_tune.h): families became bits of one key.  Tests and tools still name the members
    by the numbers they had as keys of their own; this wrapper keeps the current masks (process-wide, like the panel itself) and translates
        36, 37 -> bits 2, 4 of key 35 (35 itself: bit 1);   46, 49, 50, 55, 59, 63 -> bits 2, 4, 8, 16, 32, 64 of key 41 (41 itself: bit 1);
        47 -> the tens digit of key 31 (31 itself: the units digit).
    Every other key goes through unchanged; 52 (the AMG K-cycle) is the environment variable FV_AMG_KCYCLE now."""
    state = {35: 7, 41: 127, 31: 1}
    bits = {35: (35, 1), 36: (35, 2), 37: (35, 4), 41: (41, 1), 46: (41, 2), 49: (41, 4), 50: (41, 8), 55: (41, 16), 59: (41, 32), 63: (41, 64)}

    def tune(key, value):
        key, value = int(key), int(value)
        if key in bits:
            if value not in (0, 1):
                return FV_ERR_ARG
            k, b = bits[key]
            new = (state[k] | b) if value else (state[k] & ~b)
            rc = raw(k, new)
            if rc == 0:
                state[k] = new
            return rc
        if key in (31, 47):
            if key == 31 and not 0 <= value <= 2 or key == 47 and value not in (0, 1):
                return FV_ERR_ARG
            new = (state[31] // 10) * 10 + value if key == 31 else (1 - value) * 10 + state[31] % 10
            rc = raw(31, new)
            if rc == 0:
                state[31] = new
            return rc
        return raw(key, value)

    return tune


def load():
    """Load libfvhip.so (no GPU is touched until a context is created)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise ImportError(
            "libfvhip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C finitevolume.jl_amd/csrc`. There is no CPU fallback." % LIBPATH
        )
    lib = C.CDLL(LIBPATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(PRIVATE_SIGNATURES.items()):
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    lib.fv_tune_raw = lib.fv_tune
    lib.fv_tune = legacy_tune(lib.fv_tune_raw)
    have = lib.fv_abi_version()
    if have != ABI_VERSION:  # a stale libfvhip.so (or a newer one): the signatures above would not match
        raise ImportError("libfvhip.so speaks ABI version %d, this binding expects %d: rebuild it (make -C finitevolume.jl_amd/csrc)" % (have, ABI_VERSION))
    _lib = lib
    return lib


def ptr(a):
    """Address of a numpy array (or None)."""
    if a is None:
        return None
    return a.ctypes.data


def ai64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def af64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def check(rc, ctx=None):
    if rc == FV_OK:
        return
    lib = load()
    msg = lib.fv_last_error(ctx)
    msg = msg.decode() if msg else ""
    raise FVError(rc, msg or ("libfvhip error %d" % rc))


class Context:
    """One GPU (fv_ctx)."""

    def __init__(self, device=0):
        lib = load()
        h = c_ctx()
        rc = lib.fv_ctx_create(int(device), C.byref(h))
        if rc != FV_OK:
            check(rc, None)
        self.handle = h
        self.device = int(device)

    def check(self, rc):
        check(rc, self.handle)

    def synchronize(self):
        self.check(load().fv_ctx_synchronize(self.handle))

    def set_option(self, option, value):
        """fv_ctx_set_option: per-context options (FV_OPT_REORDER: 0 never / 1 auto / 2 always re-number face-list meshes; FV_OPT_LEAN_SETUP:
        regular-grid problems without faces and CSR in HBM — 0 never / 1 always / 2 where the CSR would not fit int32 offsets)."""
        self.check(load().fv_ctx_set_option(self.handle, int(option), int(value)))

    def get_option(self, option):
        v = C.c_int()
        self.check(load().fv_ctx_get_option(self.handle, int(option), C.byref(v)))
        return v.value

    def mem_info(self):
        """(free, total) bytes of the device."""
        f, t = C.c_int64(), C.c_int64()
        self.check(load().fv_device_mem_info(self.handle, C.byref(f), C.byref(t)))
        return f.value, t.value

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int()
        mem = C.c_int64()
        self.check(load().fv_device_info(self.handle, name, 256, C.byref(cus), C.byref(mem)))
        return name.value.decode(), cus.value, mem.value

    def close(self):
        if getattr(self, "handle", None):
            load().fv_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx
