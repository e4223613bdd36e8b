"""ctypes bindings of liblbfgsx.so / liblbfgsx_solver.so (the product path: native HIP only)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))

F64, F32 = 0, 1
OBJ_DIAG_QUAD, OBJ_EXT_ROSENBROCK = 0, 1
LS_NOCEDAL_WRIGHT, LS_MORE_THUENTE, LS_BACKTRACKING, LS_BRACKETING = 0, 1, 2, 3
ALGO_LBFGS, ALGO_LBFGSB = 0, 1
RECURSION_VECTOR, RECURSION_GRAM_SPACE, RECURSION_GRAM_SPACE_F32H = 0, 1, 2
FLAG_BOUNDED = 1
(VEC_X, VEC_G, VEC_XP, VEC_GP, VEC_D, VEC_XT, VEC_GT, VEC_A, VEC_B, VEC_LB, VEC_UB, VEC_XCP) = range(12)
E_INVALID, E_LOGIC, E_RUNTIME, E_HIP, E_NOGPU = -1, -2, -3, -4, -5


class Params(C.Structure):
    _fields_ = [("m", C.c_int), ("epsilon", C.c_double), ("epsilon_rel", C.c_double), ("past", C.c_int),
                ("delta", C.c_double), ("max_iterations", C.c_int), ("linesearch", C.c_int),
                ("max_linesearch", C.c_int), ("min_step", C.c_double), ("max_step", C.c_double),
                ("ftol", C.c_double), ("wolfe", C.c_double), ("max_submin", C.c_int)]


class Result(C.Structure):
    _fields_ = [("niter", C.c_int), ("nfev", C.c_int), ("fx", C.c_double), ("gnorm", C.c_double),
                ("status", C.c_int), ("msg", C.c_char * 200)]


class BatchItem(C.Structure):
    _fields_ = [("niter", C.c_int), ("nfev", C.c_int), ("status", C.c_int), ("fx", C.c_double), ("gnorm", C.c_double)]


class Trace(C.Structure):
    _fields_ = [("cap", C.c_int), ("count", C.c_int), ("fx", C.POINTER(C.c_double)), ("stride", C.c_int64),
                ("nsamp", C.c_int64), ("xs", C.POINTER(C.c_double))]


ITER_HOOK = C.CFUNCTYPE(None, C.c_int, C.c_void_p)
ALLREDUCE = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.c_void_p)

_core = None
_solver = None


class NativeLibraryMissing(RuntimeError):
    pass


def load():
    """Load both native libraries.  Fails loudly when they are missing: there is no fallback path."""
    global _core, _solver
    if _core is not None:
        return _core, _solver
    try:
        # When torch is (or will be) in the process its bundled HIP runtime (soname libamdhip64.so.7) must be
        # the one and only runtime, so it has to be loaded before ours resolves the same soname.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    p_core = os.path.join(HERE, "liblbfgsx.so")
    p_sol = os.path.join(HERE, "liblbfgsx_solver.so")
    for p in (p_core, p_sol):
        if not os.path.exists(p):
            raise NativeLibraryMissing(
                "%s not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')" % p)
    core = C.CDLL(p_core)
    sol = C.CDLL(p_sol)
    vp, i64, dbl, i32 = C.c_void_p, C.c_int64, C.c_double, C.c_int
    pd = C.POINTER(C.c_double)

    def sig(lib, name, res, *args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = list(args)
        return f

    sig(core, "lbfgsx_last_error", C.c_char_p)
    sig(core, "lbfgsx_version", C.c_char_p)
    sig(core, "lbfgsx_device_count", i32)
    sig(core, "lbfgsx_create", i32, C.POINTER(vp), i32, i64, i32, i32, i32)
    sig(core, "lbfgsx_destroy", None, vp)
    sig(core, "lbfgsx_set_stream", i32, vp, vp)
    sig(core, "lbfgsx_sync", i32, vp)
    sig(core, "lbfgsx_n", i64, vp)
    sig(core, "lbfgsx_vec", vp, vp, i32)
    sig(core, "lbfgsx_upload", i32, vp, i32, vp)
    sig(core, "lbfgsx_download", i32, vp, i32, vp)
    sig(core, "lbfgsx_gather", i32, vp, i32, i64, pd)
    sig(core, "lbfgsx_gen_diag_quad", i32, vp, dbl, C.c_uint64)
    sig(core, "lbfgsx_gen_rosen_x0", i32, vp, C.c_uint64)
    sig(core, "lbfgsx_fill", i32, vp, i32, dbl)
    sig(core, "lbfgsx_bfgs_reset", i32, vp)
    sig(core, "lbfgsx_bfgs_ncorr", i32, vp)
    sig(core, "lbfgsx_bfgs_theta", dbl, vp)
    sig(core, "lbfgsx_bfgs_add_correction_host", i32, vp, vp, vp)
    sig(core, "lbfgsx_bfgs_download_history", i32, vp, vp, vp, C.POINTER(i32), C.POINTER(i32), pd)
    sig(core, "lbfgsx_apply_Hv", i32, vp, i32, dbl, pd)
    sig(core, "lbfgsx_eval", i32, vp, i32, pd, pd, pd)
    sig(core, "lbfgsx_norms", i32, vp, pd, pd)
    sig(core, "lbfgsx_ls_begin", i32, vp)
    sig(core, "lbfgsx_trial", i32, vp, i32, dbl, pd, pd)
    sig(core, "lbfgsx_trial_point", i32, vp, dbl)
    sig(core, "lbfgsx_trial_dg", i32, vp, pd)
    sig(core, "lbfgsx_ls_keep_trial_as_lo", i32, vp)
    sig(core, "lbfgsx_ls_end", i32, vp, i32)
    sig(core, "lbfgsx_post_linesearch", i32, vp, pd, pd, pd, pd)
    sig(core, "lbfgsx_commit_correction", i32, vp)
    sig(core, "lbfgsx_gs_post_linesearch", i32, vp, pd, pd, pd, pd)
    sig(core, "lbfgsx_gs_set_history_dtype", i32, vp, i32)
    sig(core, "lbfgsx_gs_direction", i32, vp, pd, dbl, pd)
    sig(core, "lbfgsx_timing_enable", i32, vp, i32)
    sig(core, "lbfgsx_timing_read", i32, vp, pd, C.POINTER(i64), pd, C.POINTER(i64))
    sig(core, "lbfgsx_stream_probe", i32, vp, i32, pd, pd)
    sig(core, "lbfgsx_selftest_reduce", i32, vp, i32, i32, i32, pd)
    sig(core, "lbfgsx_post_linesearch_spec", i32, vp, dbl, pd, pd, pd, pd)
    sig(core, "lbfgsx_rccl_allgather_records", i32, C.POINTER(i32), i32, vp, i64, i64, C.POINTER(vp))
    sig(core, "lbfgsx_device_download", i32, i32, vp, i64, vp)
    sig(core, "lbfgsx_device_free", None, i32, vp)
    sig(core, "lbfgsx_spec_counts", i32, vp, C.POINTER(i64 * 3))
    sig(core, "lbfgsx_counters", i32, C.POINTER(i64 * 3), i32)
    sig(core, "lbfgsx_counters_ex", i32, C.POINTER(i64 * 8), i32)
    sig(core, "lbfgsx_poll_counts", i32, vp, C.POINTER(i64 * 2))
    sig(core, "lbfgsx_poll_counts_ex", i32, vp, C.POINTER(i64 * 4))
    sig(core, "lbfgsx_b_compact_vec_counts", i32, C.POINTER(i64 * 4), i32)
    sig(core, "lbfgsx_b_reserve", i32, vp)
    sig(core, "lbfgsx_device", i32, vp)
    sig(core, "lbfgsx_persist_counts", i32, vp, C.POINTER(i64 * 4))
    sig(core, "lbfgsx_debug_persist_fault", i32, vp)

    sig(sol, "lbfgsx_solver_create", i32, C.POINTER(vp), i32, i32, i32, C.POINTER(Params), i32)
    sig(sol, "lbfgsx_solver_create_error", C.c_char_p)
    sig(sol, "lbfgsx_solver_destroy", None, vp)
    sig(sol, "lbfgsx_solver_prepare", i32, vp, i64)
    sig(sol, "lbfgsx_solver_ctx", vp, vp)
    sig(sol, "lbfgsx_solver_set_recursion", i32, vp, i32)
    sig(sol, "lbfgsx_solver_set_allreduce", i32, vp, ALLREDUCE, vp)
    sig(sol, "lbfgsx_solver_set_devices", i32, vp, C.POINTER(i32), i32)
    sig(core, "lbfgsx_comm_unique_id", i32, C.c_char_p)
    sig(core, "lbfgsx_comm_create_rank", i32, C.POINTER(vp), i32, i32, i32, C.c_char_p)
    sig(core, "lbfgsx_comm_create_local", i32, C.POINTER(vp), C.POINTER(i32), i32)
    sig(core, "lbfgsx_comm_allreduce_sum", i32, vp, i32, pd, i32)
    sig(core, "lbfgsx_comm_abort", i32, vp)
    sig(core, "lbfgsx_comm_abort_from", i32, vp, i32)
    sig(core, "lbfgsx_comm_first_abort", i32, vp)
    sig(core, "lbfgsx_b_gram_pairs_max", i32, vp)
    sig(core, "lbfgsx_b_post_linesearch_build", i32, vp, dbl, pd, pd, pd, pd)
    sig(core, "lbfgsx_b_post_build_counts", i32, C.POINTER(i64 * 2), i32)
    sig(core, "lbfgsx_b_psel_counts", i32, C.POINTER(i64 * 1), i32)
    sig(core, "lbfgsx_b_dg_maxstep_trial", i32, vp, i32, dbl, pd, pd)
    sig(core, "lbfgsx_b_solve_sweep_rhs", i32, vp, i32, i32, pd, dbl, pd, pd, pd, C.POINTER(i64 * 7))
    sig(core, "lbfgsx_b_solve_sweep_rhs_ready", i32, vp)
    sig(core, "lbfgsx_b_gram_last_vrow_dd", i32, vp, pd)
    sig(core, "lbfgsx_b_wtv_lu_c", i32, vp, pd, C.POINTER(i64), pd, C.POINTER(i64), pd)
    sig(core, "lbfgsx_b_trial_ahead_counts", i32, vp, C.POINTER(i64 * 2))
    sig(core, "lbfgsx_comm_info", i32, vp, C.POINTER(i32 * 4))
    sig(core, "lbfgsx_comm_calls", i64, vp, i32)
    sig(core, "lbfgsx_comm_hook_arg", vp, vp, i32)
    sig(core, "lbfgsx_comm_destroy", None, vp)
    sig(core, "lbfgsx_set_shard", i32, vp, i64, i64)
    sig(sol, "lbfgsx_solver_set_iteration_hook", i32, vp, ITER_HOOK, vp)
    sig(sol, "lbfgsx_batch_minimize", i32, i32, i32, i32, C.POINTER(Params), i32, i64, i64, i64, C.c_uint64, i32, i32,
        C.POINTER(BatchItem))
    sig(sol, "lbfgsx_batch_minimize_lockstep", i32, i32, C.POINTER(Params), i64, i64, i32, C.c_uint64, i32,
        C.POINTER(BatchItem), vp, C.c_char_p, i32)
    sig(sol, "lbfgsx_batch_minimize_lockstep_multi", i32, i32, C.POINTER(Params), i64, i64, i32, C.c_uint64,
        C.POINTER(i32), i32, C.POINTER(BatchItem), vp, C.c_char_p, i32)
    sig(sol, "lbfgsx_batch_minimize_lockstep_ex", i32, i32, i32, i32, dbl, C.POINTER(Params), i64, i64, i32, C.c_uint64,
        C.POINTER(i32), i32, C.POINTER(BatchItem), vp, C.c_char_p, i32)
    sig(sol, "lbfgsx_lockstep_create", i32, C.POINTER(vp), i32, i32, C.POINTER(Params), i64, i32, i32, i32, C.c_char_p, i32)
    sig(sol, "lbfgsx_lockstep_minimize", i32, vp, i32, dbl, C.c_uint64, i64, C.POINTER(BatchItem), vp, C.POINTER(dbl * 8),
        C.c_char_p, i32)
    sig(sol, "lbfgsx_lockstep_destroy", None, vp)
    sig(sol, "lbfgsx_lockstep_set_timing", i32, vp, i32)
    sig(sol, "lbfgsx_solver_hessians", i32, vp, vp, vp)
    sig(sol, "lbfgsx_solver_stats", i32, vp, C.POINTER(C.c_longlong * 8))
    sig(sol, "lbfgsx_solver_stats2", i32, vp, C.POINTER(C.c_longlong * 8))
    sig(sol, "lbfgsx_solver_stats3", i32, vp, C.POINTER(C.c_longlong * 8))
    sig(sol, "lbfgsx_solver_minimize", i32, vp, i32, i64, vp, vp, vp, vp, vp, C.POINTER(Trace), C.POINTER(Result))
    _core, _solver = core, sol
    return core, sol


def last_error():
    core, _ = load()
    return core.lbfgsx_last_error().decode()


_EXC = {E_INVALID: ValueError, E_LOGIC: ArithmeticError, E_RUNTIME: RuntimeError, E_HIP: RuntimeError,
        E_NOGPU: RuntimeError}


def check(rc, msg=None):
    if rc != 0:
        raise _EXC.get(rc, RuntimeError)(msg if msg is not None else last_error())
