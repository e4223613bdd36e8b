/* include/lbfgsx.h -- C ABI of the MI355X-native L-BFGS / L-BFGS-B hot path (liblbfgsx.so).
 *
 * The reference (yixuan/LBFGSpp) is header-only C++ with no FFI seam; the seam this library replaces is
 * the set of BFGSMat / line-search / Cauchy / SubspaceMin call sites inside its drivers:
 *   LBFGS.h:43,91-92,121-123,127,130,137,159-165      (reset, eval, copies, line search, s/y, apply_Hv)
 *   LBFGSB.h:128-138,154,174-179,203-206,235-250      (projection, proj. gradient, max step, GCP, subspace)
 * Every function below cites the reference statement(s) it executes on the device.  The drop-in C++ API
 * (include/LBFGS.h, include/LBFGSB.h: LBFGSpp::LBFGSSolver / LBFGSBSolver / LBFGSParam) keeps all scalar
 * control flow on the host and calls only this C ABI, so user code is compiled by a plain C++ compiler.
 *
 * Conventions: plain pointers and sizes only; every call returns 0 on success or a negative LBFGSX_E_* code
 * (message via lbfgsx_last_error()); scalar results are written to caller-provided doubles (values are
 * computed in the context's scalar type and widened exactly); calls are synchronous w.r.t. the returned
 * scalars but all device work is enqueued on the context's HIP stream.  A context is single-owner and bound
 * to one device; distinct contexts are independent.
 */
#ifndef LBFGSX_H
#define LBFGSX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lbfgsx_ctx lbfgsx_ctx;

enum { LBFGSX_F64 = 0, LBFGSX_F32 = 1 };
enum { LBFGSX_OBJ_NONE = -1, LBFGSX_OBJ_DIAG_QUAD = 0, LBFGSX_OBJ_EXT_ROSENBROCK = 1 };
enum
{
    LBFGSX_OK = 0,
    LBFGSX_E_INVALID = -1, /* -> std::invalid_argument */
    LBFGSX_E_LOGIC = -2,   /* -> std::logic_error */
    LBFGSX_E_RUNTIME = -3, /* -> std::runtime_error */
    LBFGSX_E_HIP = -4,     /* HIP runtime failure (message holds hipGetErrorString) */
    LBFGSX_E_NOGPU = -5    /* no usable device: the product never falls back to a CPU path */
};
/* flags for lbfgsx_create */
enum { LBFGSX_FLAG_BOUNDED = 1 /* allocate the L-BFGS-B work set (lb, ub, xcp, masks, sort buffers) */ };

/* named device vectors of a context (length n, element type = context dtype) */
enum
{
    LBFGSX_VEC_X = 0,     /* current iterate (after a line search: the accepted point)          */
    LBFGSX_VEC_G = 1,     /* gradient at X                                                       */
    LBFGSX_VEC_XP = 2,    /* iterate at the start of the line search (m_xp, LBFGS.h:121)         */
    LBFGSX_VEC_GP = 3,    /* gradient at XP (m_gradp, LBFGS.h:122)                               */
    LBFGSX_VEC_D = 4,     /* search direction (m_drt)                                            */
    LBFGSX_VEC_XT = 5,    /* line-search trial point                                             */
    LBFGSX_VEC_GT = 6,    /* gradient at the trial point                                         */
    LBFGSX_VEC_A = 7,     /* objective data (diag quadratic: a)                                  */
    LBFGSX_VEC_B = 8,     /* objective data (diag quadratic: b)                                  */
    LBFGSX_VEC_LB = 9,    /* lower bounds (BOUNDED contexts)                                     */
    LBFGSX_VEC_UB = 10,   /* upper bounds                                                        */
    LBFGSX_VEC_XCP = 11   /* generalized Cauchy point                                            */
};

const char* lbfgsx_last_error(void);
const char* lbfgsx_version(void);
int lbfgsx_device_count(void);

/* ---- context ---------------------------------------------------------------------------------------
 * replaces LBFGSSolver::reset / BFGSMat::reset (LBFGS.h:40-50, BFGSMat.h:61-78): allocates x/g work
 * vectors and the (m+1)-column S and Y stores (column-contiguous, column stride padded to 64 elements). */
int lbfgsx_create(lbfgsx_ctx** out, int dtype, int64_t n, int m, int device, int flags);
void lbfgsx_destroy(lbfgsx_ctx* c);
int lbfgsx_set_stream(lbfgsx_ctx* c, void* hip_stream); /* adopt an external hipStream_t (e.g. torch's) */
int lbfgsx_sync(lbfgsx_ctx* c);
int64_t lbfgsx_n(const lbfgsx_ctx* c);
/* device pointer of a named vector (valid until the next call that rotates buffers) */
void* lbfgsx_vec(lbfgsx_ctx* c, int which);
int lbfgsx_upload(lbfgsx_ctx* c, int which, const void* host);   /* host -> device, n elements */
int lbfgsx_download(lbfgsx_ctx* c, int which, void* host);       /* device -> host, n elements */
int lbfgsx_gather(lbfgsx_ctx* c, int which, int64_t stride, double* host); /* host[k] = vec[k*stride] */

/* ---- synthetic problems generated on the device from a counter hash */
/* This context holds rows [offset, offset + n) of a problem of n_global rows (row-sharded runs, include/LBFGS.h
 * set_reducer): the generators below then produce exactly that slice of the unsharded data.  Default: 0, n. */
int lbfgsx_set_shard(lbfgsx_ctx* c, int64_t offset, int64_t n_global);
int lbfgsx_gen_diag_quad(lbfgsx_ctx* c, double kappa, uint64_t seed); /* fills A, B */
int lbfgsx_gen_rosen_x0(lbfgsx_ctx* c, uint64_t seed);                /* fills X    */
int lbfgsx_fill(lbfgsx_ctx* c, int which, double value);

/* ---- BFGSMat ----------------------------------------------------------------------------------------*/
/* BFGSMat::reset (BFGSMat.h:61-78): theta = 1, ncorr = 0, ptr = m */
int lbfgsx_bfgs_reset(lbfgsx_ctx* c);
int lbfgsx_bfgs_ncorr(const lbfgsx_ctx* c);
double lbfgsx_bfgs_theta(const lbfgsx_ctx* c);
/* BFGSMat::add_correction (BFGSMat.h:81-97) from host-provided s, y (testing / generic callers) */
int lbfgsx_bfgs_add_correction_host(lbfgsx_ctx* c, const void* s, const void* y);
/* same, but only stages the pair in the spare column (s.y and y.y returned); lbfgsx_commit_correction adds it */
int lbfgsx_bfgs_stage_correction_host(lbfgsx_ctx* c, const void* s, const void* y, double* sy, double* yy);
/* storage-slot order copy of the history for the dense debugging getters (BFGSMat::get_Bmat / get_Hmat,
 * BFGSMat.h:150-271): S_out, Y_out receive ncorr columns of n elements (column j = storage slot j); ptr as in
 * BFGSMat.h:42-48.  Meant for small n only. */
int lbfgsx_bfgs_download_history(lbfgsx_ctx* c, void* S_out, void* Y_out, int* ncorr, int* ptr, double* theta);
/* BFGSMat::apply_Hv (BFGSMat.h:276-302): D = a * H * v where v is a named vector; also returns
 * dg = G . D fused into the last pass when v == LBFGSX_VEC_G (LBFGS.h:123 of the next iteration). */
int lbfgsx_apply_Hv(lbfgsx_ctx* c, int v_which, double a, double* dg);

/* ---- L-BFGS driver statements ------------------------------------------------------------------------*/
/* fx = f(x, grad); gnorm = grad.norm(); x.norm()   (LBFGS.h:91-92,100) with a built-in objective */
int lbfgsx_eval(lbfgsx_ctx* c, int objective, double* fx, double* gnorm2, double* xnorm2);
/* reductions only, for user (device-functor) objectives that filled G themselves */
int lbfgsx_norms(lbfgsx_ctx* c, double* gnorm2, double* xnorm2);
/* xp = x; gradp = grad (LBFGS.h:121-122) by buffer rotation; x_lo/grad_lo alias them
 * (LineSearchMoreThuente.h:393, LineSearchNocedalWright.h:128) */
int lbfgsx_ls_begin(lbfgsx_ctx* c);
/* x = xp + step*drt; fx = f(x, grad); dg = grad.dot(drt)
 * (LineSearchMoreThuente.h:412-414; LineSearchNocedalWright.h:146-148,219-221) */
int lbfgsx_trial(lbfgsx_ctx* c, int objective, double step, double* fx, double* dg);
/* device-functor path: XT = xp + step*drt only; then the caller fills GT; then lbfgsx_trial_dg */
int lbfgsx_trial_point(lbfgsx_ctx* c, double step);
int lbfgsx_trial_dg(lbfgsx_ctx* c, double* dg);
/* x_lo.swap(x); grad_lo.swap(grad)  (LineSearchMoreThuente.h:534-535,553-554; NocedalWright.h:172-173,254-255) */
int lbfgsx_ls_keep_trial_as_lo(lbfgsx_ctx* c);
/* end of the search: the accepted point is the last trial (use_lo = 0) or the saved _lo point
 * (use_lo = 1: LineSearchMoreThuente.h:612-613, NocedalWright.h:191-192,274-275) */
int lbfgsx_ls_end(lbfgsx_ctx* c, int use_lo);
/* gnorm, x.norm(), s = x - xp, y = grad - gradp, s.y, y.y  (LBFGS.h:130,137,159-161) in one pass;
 * s and y land in the spare history column */
int lbfgsx_post_linesearch(lbfgsx_ctx* c, double* gnorm2, double* xnorm2, double* sy, double* yy);
/* BFGSMat::add_correction of the pair just formed (BFGSMat.h:81-97): index rotation only */
int lbfgsx_commit_correction(lbfgsx_ctx* c);
/* lbfgsx_post_linesearch with the next statement of the driver folded in when the pair turns out usable:
 *     s, y, the four sums (LBFGS.h:130,137,159-161)   and, speculatively,   drt = a * H * grad  (LBFGS.h:165)
 * for the history "stored pairs + the pair just formed" -- as step 0 and steps 1..2c of ONE persistent launch: the first
 * step of the recursion needs exactly grad and the new s, which the post pass holds in registers (2n elements fewer per
 * iteration, one launch less).  The kernel applies the driver's test s.y > eps y.y (LBFGS.h:161) itself and stops after
 * the post statements when it fails.  The caller proceeds exactly as after lbfgsx_post_linesearch: convergence tests,
 * lbfgsx_commit_correction if the pair is accepted, lbfgsx_apply_Hv(LBFGSX_VEC_G, a) -- which returns the direction
 * already computed when the speculation applies (same history, same gradient, same a) and computes it otherwise.
 * Falls back to lbfgsx_post_linesearch when the persistent kernel is unavailable (LBFGSX_PERSIST=0, LBFGSX_FUSE_POST=0,
 * m > 128, another persistent launch in flight on the device).  Bit-identical results either way. */
int lbfgsx_post_linesearch_spec(lbfgsx_ctx* c, double a, double* gnorm2, double* xnorm2, double* sy, double* yy);
/* instrumentation: {fused launches, directions taken over by lbfgsx_apply_Hv, pairs rejected by the kernel} */
int lbfgsx_spec_counts(const lbfgsx_ctx* c, int64_t out[3]);
/* Every entry point of this ABI makes the context's device current for its own duration and restores the caller's
 * afterwards.  Code that launches its OWN kernels on the context's vectors (a device functor, lbfgsx_vec) must run with
 * that device current too: lbfgsx_device tells which one it is, lbfgsx_device_push makes it current for the calling
 * thread (returning the previous one in *prev), lbfgsx_device_pop restores.  The drop-in solvers bracket every call of a
 * device functor with the pair (LBFGSpp/Device.h, Evaluator::call_user). */
int lbfgsx_device(const lbfgsx_ctx* c);
int lbfgsx_device_push(const lbfgsx_ctx* c, int* prev);
int lbfgsx_device_pop(int prev);
/* The persistent launch survives a device it does not own: a meeting point that waits longer than 100 ms flags the launch,
 * the product is redone with the 2c+1 step launches (bit-identical), which the context keeps for 8 products before it tries
 * the persistent form again; every further time-out quadruples that pause, a clean persistent launch resets it.
 * out = {persistent launches, time-outs, products left of the current pause, products computed with step launches}. */
int lbfgsx_persist_counts(const lbfgsx_ctx* c, int64_t out[4]);
/* test hook: sets the failure word the next persistent launch of this context will find (as if a meeting point had timed
 * out); that launch does nothing and the host takes the recovery path above */
int lbfgsx_debug_persist_fault(lbfgsx_ctx* c);
/* instrumentation, process-wide: {kernel launches, stream synchronisations, asynchronous copies} issued by the library
 * since load (or the last call with reset != 0).  bench.py divides them by the iterations of its L-BFGS-B leg. */
int lbfgsx_counters(int64_t out[3], int reset);
/* the same three and the byte model of the L-BFGS-B path as built: out[3] = bytes the launches of the path had to move for
 * the rows and columns they were launched over (columns x rows of the compact copy of the free rows + the vectors each
 * pass reads and writes per row, gathers at 64-byte sectors; DESIGN.md section 5 lists the per-kernel terms), out[4] =
 * passes over the compact copy, out[5] = rows those passes walked, out[6..7] = 0.  bench.py's cfg4 roofline fraction is
 * out[3] of its window / time / HBM peak.  No reference counterpart (instrumentation). */
int lbfgsx_counters_ex(int64_t out[8], int reset);
/* Polled completion (contexts with mapped outputs, i.e. L-BFGS-B ones; LBFGSX_POLL=0 switches it off): the last block of a
 * kernel whose results the host reads next stores a sequence number in host-mapped memory after the results, and the host
 * polls that word instead of waiting for the stream (which returns ~9 us after the kernel's end).  out = {waits served by
 * polling, waits that timed out (50 ms) and fell back to the stream wait -- a kernel that failed to signal} */
int lbfgsx_poll_counts(const lbfgsx_ctx* c, int64_t out[2]);
/* The same with the reasons polling was given up: out = {waits, time-outs, time-outs that said "polling cannot work here"
 * (the word still unset after the stream drained, or the stream wait returned at once because the kernel had ended long
 * before its store became visible), 1 if this context now waits for its stream instead of polling (after two of those)} */
int lbfgsx_poll_counts_ex(const lbfgsx_ctx* c, int64_t out[4]);
/* Tracing: the drop-in solvers bracket their phases -- iteration, line search, post + apply_Hv, the
 * generalized Cauchy point, the subspace minimisation, a lock-step iteration of the batch -- with these calls.  With
 * LBFGSX_ROCTX=1 they become ROCTx ranges (roctxRangePushA / roctxRangePop, the marker library loaded with dlopen), which
 * `rocprofv3 --marker-trace --kernel-trace` shows beside the kernels; otherwise they cost one branch.  `name` must be a string
 * with static storage duration.  No reference counterpart (the reference has no tracing). */
void lbfgsx_range_push(const char* name);
void lbfgsx_range_pop(void);

/* ---- Gram-space ("vector-free") form of the recursion: opt-in, outside the bit-parity contract ----
 * BFGSMat::apply_Hv (BFGSMat.h:276-302) only combines the 2c+1 vectors [S, Y, g]; with their Gram matrix kept on the
 * host (include/LBFGSpp/GramSpace.h) an iteration needs two passes over the history instead of 2c+1 dependent ones.
 * Supported for m <= 24 (LBFGSX_E_INVALID otherwise). */
/* lbfgsx_post_linesearch (LBFGS.h:130,137,159-161; s, y into the spare column) plus, in the same pass, the Gram rows
 * of the new pair and of the new gradient.  scal = {g.g, x.x, s.y, y.y, s.s, g.s, g.y}; for the logical slots
 * j < lbfgsx_bfgs_ncorr(): sdots[j] = S_j.s, sdots[m+j] = Y_j.s, gdots[j] = S_j.g, gdots[m+j] = Y_j.g (arrays of 2m).
 * Sums accumulate in f64 (fixed reduction order).  ydots (2m doubles, may be NULL) receives S_j.y, Y_j.y when the
 * history is kept in f32 (below) -- there the dots are taken from the rounded, stored y; with a native history the caller
 * derives them as differences of the gradient dots and ydots is left untouched. */
int lbfgsx_gs_post_linesearch(lbfgsx_ctx* c, double scal[7], double* sdots, double* gdots, double* ydots);
/* Mixed-precision history: dtype = LBFGSX_F32 on an f64 context makes the Gram-space entry points keep
 * S and Y as float (half the history traffic and memory; x, g, d and all sums stay f64).  Only with an empty history
 * (after lbfgsx_bfgs_reset); dtype = the context's own type switches back.  While it is on, the vector-form entry points
 * that touch the history (lbfgsx_apply_Hv with pairs stored, lbfgsx_post_linesearch, the history getters) return
 * LBFGSX_E_LOGIC. */
int lbfgsx_gs_set_history_dtype(lbfgsx_ctx* c, int dtype);
/* D = coef_g * G + sum_{j < ncorr} coef[j] * S_j + coef[m+j] * Y_j (logical slots); *dg = G . D  (LBFGS.h:123) */
int lbfgsx_gs_direction(lbfgsx_ctx* c, const double* coef, double coef_g, double* dg);

/* ---- L-BFGS-B device operators (contexts created with LBFGSX_FLAG_BOUNDED) ------------------------------
 * Index sets of the reference (fv_set, newact_set, BOXCQP's L/U/P: std::vector<int>) are a per-coordinate
 * state byte on the device; every operator streams the column-contiguous S/Y once and applies the mask. */
enum /* state-byte masks */
{
    LBFGSX_ST_FREE = 1, LBFGSX_ST_NEWACT = 2, LBFGSX_ST_L = 4, LBFGSX_ST_U = 8, LBFGSX_ST_P = 16
};
enum /* vectors formed on the fly inside masked operators */
{
    LBFGSX_VS_DRT = 0,      /* drt = xcp - x0                 (SubspaceMin.h:130)   */
    LBFGSX_VS_NEG_CF = 1,   /* -vecc                          (SubspaceMin.h:159)   */
    LBFGSX_VS_NEG_RHS = 2,  /* -rhs                           (SubspaceMin.h:243)   */
    LBFGSX_VS_LBOUND = 3,   /* vecl = lb - x0                 (SubspaceMin.h:153)   */
    LBFGSX_VS_UBOUND = 4,   /* vecu = ub - x0                 (SubspaceMin.h:154)   */
    LBFGSX_VS_Y = 5         /* vecy                                                   */
};
enum /* lbfgsx_b_wcombine modes: element-wise epilogue around (W_mask * coef)(i) */
{
    LBFGSX_CB_LINEAR = 0,   /* vecc = -W_F M (W'AA'd) + g_F   (BFGSMat.h:521 + SubspaceMin.h:155)        */
    LBFGSX_CB_SOLVE = 1,    /* vecy = v/theta + W_P coef/theta^2   (BFGSMat.h:535,564)                    */
    LBFGSX_CB_RHS_ADD = 2,  /* rhs += -W_P coef               (BFGSMat.h:592 + SubspaceMin.h:236-241)    */
    LBFGSX_CB_LAMBDA = 3,   /* lambda_L = -W_L coef + vecc_L + theta*vecy_L   (SubspaceMin.h:256-258)    */
    LBFGSX_CB_MU = 4        /* mu_U = -( -W_U coef + vecc_U + theta*vecy_U )  (SubspaceMin.h:265-267)    */
};
enum /* lbfgsx_b_sub_op */
{
    LBFGSX_SO_SAVE_FALLBACK = 0, LBFGSX_SO_RHS_INIT = 1, LBFGSX_SO_ASSIGN_Y = 2, LBFGSX_SO_CLAMP_Y = 3,
    LBFGSX_SO_CLAMP_FB = 4, LBFGSX_SO_ASSIGN_FB = 5
};
/* force_bounds: x = x.cwiseMax(lb).cwiseMin(ub)  (LBFGSB.h:55-58,128,240) */
int lbfgsx_b_force_bounds(lbfgsx_ctx* c);
/* the same statement where the generalized-Cauchy-point build follows at once (LBFGSB.h:240-241): nothing is launched,
 * the build's own pass -- which reads x, lb and ub anyway -- clamps on the way (a coordinate already inside its bounds
 * costs no store).  Any other entry of the bounded path called in between runs the statement first.  Same x, same bits. */
int lbfgsx_b_force_bounds_deferred(lbfgsx_ctx* c);
/* fx = f(x,grad); ||P(x-g,l,u)-x||_inf; x.x   (LBFGSB.h:137-138,146) */
int lbfgsx_b_eval(lbfgsx_ctx* c, int objective, double* fx, double* projgnorm, double* xnorm2);
/* the two reductions alone, for objectives evaluated by the caller (device / host functors) */
int lbfgsx_b_norms(lbfgsx_ctx* c, double* projgnorm, double* xnorm2);
/* dg = grad.dot(drt); step_max = max_step_size(x,drt,lb,ub)   (LBFGSB.h:68-86,176-179,195-196) */
int lbfgsx_b_dg_maxstep(lbfgsx_ctx* c, double* dg, double* step_max);
/* the same, and in the same pass the line search's first trial at step0 (the caller's min(1, max_step): LBFGSB.h:200-203 start
 * the search at min(1, step_max)): x_trial = xp + step0 * drt, f and grad there, grad.drt (LineSearchMoreThuente.h:261-262) --
 * kept inside the context and handed to the next lbfgsx_trial() iff it asks for this objective at exactly this step; any other
 * call drops them.  Built-in objectives on contexts with mapped outputs; everything else, LBFGSX_TRIAL_AHEAD=0 and the four
 * iterations after an unused trial behave as lbfgsx_b_dg_maxstep.  To be called right after lbfgsx_ls_begin. */
int lbfgsx_b_dg_maxstep_trial(lbfgsx_ctx* c, int objective, double step0, double* dg, double* step_max);
/* instrumentation: {trials evaluated ahead, of which the line search used} */
int lbfgsx_b_trial_ahead_counts(const lbfgsx_ctx* c, int64_t out[2]);
/* after the line search: proj_grad_norm, x.x, s, y (into the spare column), s.y, y.y  (LBFGSB.h:206,213,235-237) */
int lbfgsx_b_post_linesearch(lbfgsx_ctx* c, double* projgnorm, double* xnorm2, double* sy, double* yy);
/* the same, and in the same pass over x and g the element-wise part of the Cauchy search that follows when the iteration goes
 * on (Cauchy.h:95,111-129; what lbfgsx_b_cauchy_build_partial(c, tau, ...) launches first): that call then finds its
 * break points, d, xcp = x0 and counts ready -- provided nothing has moved x since, the threshold is the same and the
 * deferred x = clamp(x) (LBFGSB.h:240) has nothing to move; otherwise it runs its own pass.  LBFGSX_POST_BUILD=0, contexts
 * without mapped outputs and the integer Gram keep the two passes.  (Replaces nothing new in the reference: LBFGSB.h:206-241
 * in one sweep over the vectors instead of two.) */
int lbfgsx_b_post_linesearch_build(lbfgsx_ctx* c, double tau, double* projgnorm, double* xnorm2, double* sy, double* yy);
/* instrumentation, process-wide: {passes of lbfgsx_b_post_linesearch_build that carried the Cauchy half, Cauchy searches that
 * used it} */
int lbfgsx_b_post_build_counts(int64_t out[2], int reset);
/* Instrumentation of the short partial sort (round 5; no reference counterpart): out[0] = candidate lists of <= 4096 rows
 * (the break points below the search's threshold, Cauchy.h:183-199) ordered by one block (k_psel_sort_small) instead of three
 * launches.  LBFGSX_PSEL_SMALL=0 switches the one-block form off. */
int lbfgsx_b_psel_counts(int64_t out[1], int reset);
/* add_correction tail: sdots[j] = S_j.s_new, ydots[j] = Y_j.s_new for slots j < ncorr  (BFGSMat.h:111,138) */
int lbfgsx_b_correction_dots(lbfgsx_ctx* c, double* sdots, double* ydots);
/* ask the next lbfgsx_b_cauchy_build* to take those dots in the pass that computes W'd (Cauchy.h:152) -- the same 2c
 * columns, one read instead of two; lbfgsx_b_correction_dots then returns them without a launch.  Same sums, same
 * bits.  Without a following build (or with 4c > 40 reductions) lbfgsx_b_correction_dots computes them as before. */
int lbfgsx_b_correction_dots_defer(lbfgsx_ctx* c);
/* GCP build: break points, vecd, xcp = x0, radix sort of the finite positive break points; returns the counts
 * of free (brk = inf) and ordered coordinates, d.d, and the raw W'd dots [Y'd, S'd]  (Cauchy.h:93-133,152-154) */
int lbfgsx_b_cauchy_build(lbfgsx_ctx* c, int64_t* nfree, int64_t* nord, double* dd, double* wtd);
/* sorted break points [first, first+count): brk, g, z = bound - x0, index, W row [y_0..y_{c-1}, s_0..s_{c-1}]
 * for the sequential piecewise-quadratic scan kept on the host (Cauchy.h:183-256) */
int lbfgsx_b_cauchy_chunk(lbfgsx_ctx* c, int64_t first, int64_t count, double* brk, double* g, double* z, int* idx,
                          double* wrows);
/* lbfgsx_b_cauchy_build with a PARTIAL sort: only the candidates whose break point is <= tau are compacted (in
 * index order) and sorted -- in steady state a search crosses 10^2..10^3 of the ~n/2 candidates, so sorting all of
 * them (8 radix passes over n pairs) is the largest item of the build.  *nsorted = length of the sorted prefix
 * (== *nord when tau <= 0 or not finite: full sort).  Every break point <= tau is in the prefix, so the search is
 * exact as long as it stops before tau; the caller falls back to lbfgsx_b_cauchy_sort_full otherwise. */
int lbfgsx_b_cauchy_build_partial(lbfgsx_ctx* c, double tau, int64_t* nfree, int64_t* nord, int64_t* nsorted, double* dd,
                                  double* wtd);
int lbfgsx_b_cauchy_sort_full(lbfgsx_ctx* c);
/* Device form of the break-point search (reference Cauchy.h:183-256) over sorted positions [first, first+count) of
 * the list produced by lbfgsx_b_cauchy_build: three dependent prefix sums (p; c and f''; f') and a min-index exit
 * test, see lbfgspp_amd/csrc/gcp_scan.cuh.  Mmat: explicit 2c x 2c matrix of apply_Mv (BFGSMat.h:361-376), column
 * major; state_in = [p (2c), c (2c), f', f'']; t_prev = break point of the last crossing already processed (0 at
 * the start).  On return *exit_at = sorted index of the group end at which the search stops (-1: not inside this
 * range) and state_out = [p, c, f', f'', brk] after that crossing (after the last crossing of the range when -1).
 * 2c <= 80 (every m an L-BFGS-B context accepts); f32 problems are gathered into doubles and searched in double.
 * The order-sensitive scalar recurrences f', f'' and the exit test run in the reference's left-to-right order on the
 * host over per-crossing terms produced by the device (LBFGSX_GCP_CHAIN=scan: as tree-order prefix sums on the device). */
int lbfgsx_b_cauchy_scan(lbfgsx_ctx* ctx, int64_t first, int64_t count, int64_t nord, const double* Mmat, double theta,
                         double t_prev, const double* state_in, int64_t* exit_at, double* state_out);
/* xcp and the free / newly-active sets from the crossing threshold (Cauchy.h:201-206,219-233,265-282) */
int lbfgsx_b_cauchy_finish(lbfgsx_ctx* c, double t_cross, double tfinal, int crossed_all, int64_t* nact, int64_t* nfree);
/* drt = xcp - x0 (SubspaceMin.h:130) */
int lbfgsx_b_sub_begin(lbfgsx_ctx* c);
/* raw masked W'v: out[0..c) = Y_j.v, out[c..2c) = S_j.v over coordinates whose state has `mask` bits
 * (apply_Wtv / apply_WtPv, BFGSMat.h:315-320,382-430); nnz = non-zero entries of v inside the mask */
int lbfgsx_b_wtv(lbfgsx_ctx* c, int vsel, int mask, double* out, int64_t* nnz);
/* masked Gram of [Y_P, S_P] (2c x 2c, row-major, symmetric) for solve_PtBP (BFGSMat.h:543-556) */
int lbfgsx_b_gram(lbfgsx_ctx* c, int mask, double* gram);
/* the same Gram plus W_P'v (v = on-the-fly vector `vsel`, or -1) in ONE pass over the 2c columns: correctly rounded
 * double-double sums (the default kernels), or the exact integer-MFMA form with LBFGSX_GRAM=i8.  (Rounds 1-3 had a plain
 * f64 MFMA form behind this entry, LBFGSX_GRAM=mfma: ~1 ulp per entry, not exact, removed in round 4.)  Returns
 * LBFGSX_E_INVALID when the one-pass form does not apply -- callers then use lbfgsx_b_gram + lbfgsx_b_wtv. */
int lbfgsx_b_gram_fused(lbfgsx_ctx* c, int mask, int vsel, double* gram, double* wtv);
/* The same with the combine statement that produces v fused in as a prologue (one pass instead of two or three):
 *   LBFGSX_GP_RHS     rhs = (rhs + -(W_mask coef1)) + -(W_mask coef2), then v = -rhs   (two apply_PtBQv, BFGSMat.h:570-594;
 *                     a NULL coefficient vector skips its term)
 *   LBFGSX_GP_LINEAR  cF = -1 * (W_mask coef1) + g (coef1 NULL: cF = g), then v = -cF (compute_FtBAb, BFGSMat.h:486-522)
 * coef*: 2c doubles, logical order [Y slots, S slots] as for lbfgsx_b_wcombine.  Default Gram kernel only. */
enum { LBFGSX_GP_NONE = 0, LBFGSX_GP_RHS = 1, LBFGSX_GP_LINEAR = 2 };
int lbfgsx_b_gram_fused_ex(lbfgsx_ctx* c, int mask, int vsel_id, int prologue, const double* coef1, const double* coef2,
                           double* gram, double* wtv);
/* The same pass that additionally returns the UN-ROUNDED double-double sums of the 2c x 2c block: gram_dd[2 e],
 * gram_dd[2 e + 1] = (hi, lo) of entry e = i (i + 1) / 2 + j (i >= j), 2c (2c + 1) doubles.  They let the caller form
 * the Gram of a subset through the complement identity  W_P'W_P = W_F'W_F - W_{F\P}'W_{F\P}: both sums are accurate to
 * ~2^-100, so their difference rounds to the same double as the direct sum -- and in a BOXCQP sweep the complement
 * L u U holds 10^1..10^3 rows against |P| ~ n/2.  gram and wtv may be NULL.  Default Gram kernel only. */
int lbfgsx_b_gram_fused_dd(lbfgsx_ctx* c, int mask, int vsel_id, int prologue, const double* coef1, const double* coef2,
                           double* gram, double* wtv, double* gram_dd);
/* The v row of lbfgsx_b_gram_fused_ex alone: the prologue statement on the rows of `mask`, then the raw masked
 * multi-dot wtv = [Y_mask'v, S_mask'v] -- the Gram kernel with one pair per lane instead of all of them. 1 <= 2c <= 30. */
int lbfgsx_b_wtv_prologue(lbfgsx_ctx* c, int mask, int vsel_id, int prologue, const double* coef1, const double* coef2,
                          double* wtv);
/* lbfgsx_b_wcombine(LBFGSX_CB_SOLVE, pmask, vsel, coef, theta) fused with the raw masked multi-dot
 * wty = [Y_F' y, S_F' y] over `fmask` (pmask must be a subset of fmask): the solve result of solve_PtBP
 * (BFGSMat.h:564) and the W_F' y the multipliers need (SubspaceMin.h:249-254, apply_WtPv BFGSMat.h:382-430) in one
 * pass.  coef may be NULL (y = v/theta).  1 <= 2c <= 32, LBFGSX_E_INVALID otherwise. */
int lbfgsx_b_solve_wty(lbfgsx_ctx* c, int pmask, int vsel, const double* coef, double theta, int fmask, double* wty);
/* masked combine with element-wise epilogue, see LBFGSX_CB_*; coef = NULL means "W term absent" */
int lbfgsx_b_wcombine(lbfgsx_ctx* c, int mode, int mask, int vsel, const double* coef, double theta);
/* BOXCQP partition of the free set into L/U/P with the value/multiplier updates (SubspaceMin.h:194-219) */
int lbfgsx_b_sub_partition(lbfgsx_ctx* c, int64_t* nL, int64_t* nU, int64_t* nP);
/* counts = {#F outside [l,u], #P outside [l,u], #L with lambda<0, #U with mu<0}  (SubspaceMin.h:60-108) */
int lbfgsx_b_sub_check(lbfgsx_ctx* c, int64_t counts[4]);
/* the element-wise statements between two BOXCQP solves in ONE pass: first != 0: LBFGSX_SO_SAVE_FALLBACK
 * (SubspaceMin.h:170-172) with counts[0] = #F outside [l,u] (the in_bounds test of the first solve, :162; when it is 0
 * the pass has moved no value of y), else the counts[1..3] of lbfgsx_b_sub_check on the current values; then
 * lbfgsx_b_sub_partition (:194-219) and LBFGSX_SO_RHS_INIT (:232).  Bit for bit the calls it replaces. */
int lbfgsx_b_sub_sweep_begin(lbfgsx_ctx* c, int first, int64_t* nL, int64_t* nU, int64_t* nP, int64_t counts[4]);
/* lbfgsx_b_wtv(LBFGSX_VS_LBOUND, LBFGSX_ST_L) and lbfgsx_b_wtv(LBFGSX_VS_UBOUND, LBFGSX_ST_U) -- the inner products of the two
 * apply_PtBQv statements of a BOXCQP sweep (SubspaceMin.h:236-241, BFGSMat.h:570-594) -- in one launch over the index list of
 * L u U.  LBFGSX_E_INVALID when there is no list or 2c > 24: call lbfgsx_b_wtv per set. */
int lbfgsx_b_wtv_lu(lbfgsx_ctx* c, double* out_l, int64_t* nnz_l, double* out_u, int64_t* nnz_u);
/* the same, plus negc_dd[2 k], [2 k + 1] = the un-rounded (hi, lo) sum of column k of W_{L u U}'(-c) (c = vecc of
 * SubspaceMin.h on the rows of L u U); negc_dd[0] = NaN when this context cannot deliver it (LBFGSX_RHS_IDENTITY=0, no
 * split-row kernels, no mapped outputs) */
int lbfgsx_b_wtv_lu_c(lbfgsx_ctx* c, double* out_l, int64_t* nnz_l, double* out_u, int64_t* nnz_u, double* negc_dd);
/* ---- pieces of the carried Gram of the free set (BFGSMatB::solve_PtBP): W_F'W_F of one iteration from that of the one
 * before -- the entries of the columns add_correction replaced computed afresh, the others corrected by the outer products
 * of the rows that entered or left F.  All sums un-rounded double-doubles (hi, lo).
 *
 * lbfgsx_b_free_delta: the rows whose membership of F (LBFGSX_ST_FREE) differs from what it was at the previous call (the
 * first call: from the empty set) are put on two lists, their sizes returned (-1: more rows than the list
 * holds -- n / 64 of them, at least 2^14 and at most 2^20 -- so it is not usable); the remembered set becomes the current one. */
int lbfgsx_b_free_delta(lbfgsx_ctx* c, int64_t* n_enter, int64_t* n_leave);
/* the 2c x 2c Gram of [Y S] over the rows of one of those lists (0: entered, 1: left), packed lower triangle
 * e = i (i + 1) / 2 + j as lbfgsx_b_gram_fused_dd returns it; LBFGSX_E_INVALID for an empty or overflowed list */
int lbfgsx_b_gram_list_dd(lbfgsx_ctx* c, int which, double* gram_dd);
/* selected entries of [Y_P S_P v]'[Y_P S_P v] in one pass with the prologue of lbfgsx_b_gram_fused_ex: entry e is the
 * product of the columns pair_i[e], pair_j[e] (0..2c-1: Y slots then S slots; 2c: v), at most lbfgsx_b_gram_pairs_max()
 * of them; out_dd[2 e], out_dd[2 e + 1] = (hi, lo).  Writes the compact copy of the free rows
 * under the conditions of lbfgsx_b_set_compaction. */
int lbfgsx_b_gram_pairs_dd(lbfgsx_ctx* c, int mask, int vsel, int prologue, const double* coef1, const double* coef2,
                           int npairs, const int* pair_i, const int* pair_j, int refresh_slot, double* out_dd);
/* how many entries one lbfgsx_b_gram_pairs_dd call serves with the current history: 3 (2c + 1) -- the v row and the rows of
 * two columns, whatever 2c <= 80 is (kernels of csrc/lbfgsb_x.cuh) -- or, with LBFGSX_SPLIT=0, the 64 of the round-3
 * one-entry-per-lane kernel while 2c + 1 <= 31; 0: the call is not available for this history */
int lbfgsx_b_gram_pairs_max(lbfgsx_ctx* c);
/* the un-rounded (hi, lo) sums of the v row W_P'v of the one-pass Gram this context ran last (lbfgsx_b_gram_fused_dd with a
 * vector selector, not a list): out_dd[2 k], [2 k + 1] for column k < 2c.
 * LBFGSX_E_INVALID when there is none.  BFGSMatB::solve_PtBP keeps W_F'(-c) of a first solve that took the full pass, so
 * that its sweeps can form W_P' rhs on the host (see lbfgsx_b_solve_sweep_rhs). */
int lbfgsx_b_gram_last_vrow_dd(lbfgsx_ctx* c, double* out_dd);
/* refresh_slot >= -1: the caller vouches that since the previous subspace minimisation the history changed in at most the
 * storage slot `refresh_slot` (-1: not at all) and that lbfgsx_b_free_delta has been called for the current free set.  The
 * pass may then read the compact copy of the free rows it KEPT from that minimisation (rows that entered F were appended by
 * lbfgsx_b_free_delta, rows that left are skipped through their state byte) and only rewrites the two columns of that slot
 * in it, instead of reading every row of the 2c columns through the mask and writing the copy afresh.  -2: never. */
/* Hint for the subspace minimisation that lbfgsx_b_sub_begin has just opened (which clears it): BOXCQP sweeps are expected,
 * so the full Gram pass of the first solve (lbfgsx_b_gram_fused_dd over LBFGSX_ST_FREE) may also write a compact copy of the
 * free rows of [Y S], and the passes of the sweeps (lbfgsx_b_wtv_prologue, lbfgsx_b_solve_sweep, the complement Grams) then
 * read that copy instead of fetching all n rows through the state-byte mask.  Same sums (double-double), same bits; only
 * taken when the free set leaves out at least an eighth of the rows.  LBFGSX_COMPACT_FREE=0 disables it.  The copy is
 * valid until the next lbfgsx_b_sub_begin / lbfgsx_b_cauchy_finish; the history must not change in between. */
int lbfgsx_b_set_compaction(lbfgsx_ctx* c, int enable);
/* Compact vectors (round 3).  While lbfgsx_b_solve_sweep / _lu_sweep / _wtv_lu / _wtv_prologue and the L u U complement of
 * lbfgsx_b_gram_fused_dd sweep over the compact copy, vecy, yfallback, lambda, mu, rhs, c_F, l - x0, u - x0 and the
 * partition bits of the free rows (SubspaceMin.h:159-268) sit at the rows' POSITIONS in the copy (contiguous) instead of at
 * the rows; lbfgsx_b_sub_op(LBFGSX_SO_ASSIGN_Y) assigns the result from there, any other entry of the bounded path first
 * puts them back at their rows.  Same statements on the same values: no bit changes.  LBFGSX_COMPACT_VEC=0 keeps the
 * vectors at their rows.
 * The kept copy also serves the Cauchy search (round 3): p = W'd (Cauchy.h:152) and the deferred dots of add_correction
 * (BFGSMat.h:111,138) are sums over the rows where d or s_new is not zero -- the positions of the copy plus a short list of
 * other rows that lbfgsx_b_cauchy_build* writes on its way -- instead of over all n rows (LBFGSX_WTD_COMPACT=0: all rows).
 * Instrumentation, process-wide: out = {minimisations that ran on compact vectors, times they were put back before the
 * result was assigned, Cauchy searches whose W'd came from the copy, Grams over index lists that were launched behind the
 * pass before them and cost no round trip of their own (LBFGSX_SYNC_MERGE=0: none)}. */
int lbfgsx_b_compact_vec_counts(int64_t out[4], int reset);
/* Setup: allocates now what the bounded path allocates on first use -- the buffers of the device break-point search, of the
 * partial sort and of the free-set delta, the compact copy of the free rows and its vectors -- so that a solve on a prepared
 * context (LBFGSBSolver::prepare_resident) contains no allocation.  Optional: without it the first iterations allocate
 * (a few milliseconds at n = 10^7).  The optional work sets are skipped silently when there is no room for them. */
int lbfgsx_b_reserve(lbfgsx_ctx* c);
/* A BOXCQP solve and the statements of lbfgsx_b_sub_sweep_begin on the rows it writes, in ONE pass (the solve's row of W is
 * in registers; the sweep's pass over n rows disappears).  Bit for bit lbfgsx_b_wcombine(LBFGSX_CB_SOLVE) /
 * lbfgsx_b_solve_wty followed by lbfgsx_b_sub_sweep_begin.
 *   first != 0: y = v/theta + (W coef)/theta^2 on every free row (SubspaceMin.h:159), then the first sweep's statements;
 *               sums = {#L, #U, #P, #F outside, 0, 0, 0}; wty is not written.
 *   first == 0: y on the rows of P (:243) and wty = W_F'y as lbfgsx_b_solve_wty, then the next sweep's statements on the
 *               rows of P (their multipliers are zero); sums = {#L, #U, #P, 0, #P outside, 0, 0} over those rows.  The
 *               rows of the old L and U follow in lbfgsx_b_lu_sweep, which needs the multipliers' coefficients the caller
 *               derives from wty; the totals are the sums of the two calls.  Needs the index list of L u U that the
 *               previous sweep kept (small sets).
 * vsel: the selector of the ONE vector v solved for (LBFGSX_VS_NEG_CF, LBFGSX_VS_NEG_RHS, ...); LBFGSX_VS_LBOUND / _UBOUND
 * (the bound vectors of lbfgsx_b_wtv_lu) are refused.
 * LBFGSX_E_INVALID when the fused form does not apply here (2c > 80, no list, LBFGSX_SWEEP_SOLVE_FUSE=0, a bound selector):
 * nothing has been changed, run the separate calls.
 * After LBFGSX_SO_ASSIGN_Y has run on live compact vectors the partition bits (LBFGSX_ST_P / _L / _U) of the state byte
 * are not written back (only the free / active bits are read afterwards): lbfgsx_b_download_state returns them undefined. */
int lbfgsx_b_solve_sweep(lbfgsx_ctx* c, int first, int vsel, const double* coef, double theta, double* wty, int64_t sums[7]);
/* the same with the two rhs updates that precede a sweep's solve (SubspaceMin.h:236-241; the LBFGSX_GP_RHS prologue of
 * lbfgsx_b_wtv_prologue with these coefficients) evaluated by the solve's own pass on the W rows it holds: first = 0 and
 * vsel = LBFGSX_VS_NEG_RHS only, split-row kernels only.  The caller then needs W_P'(-rhs) from elsewhere -- BFGSMatB::solve_PtBP
 * forms it from sums it holds (W_F'(-c), W_{L u U}'(-c) of lbfgsx_b_wtv_lu_c, the Gram of the complement identity) -- and the pass
 * over P that lbfgsx_b_wtv_prologue makes for it is saved.  _ready: 1 when this context can do it now. */
int lbfgsx_b_solve_sweep_rhs(lbfgsx_ctx* c, int first, int vsel, const double* coef, double theta, const double* rhs_c1,
                             const double* rhs_c2, double* wty, int64_t sums[7]);
int lbfgsx_b_solve_sweep_rhs_ready(lbfgsx_ctx* c);
/* completes lbfgsx_b_solve_sweep(first == 0): lambda on the rows of L, mu on the rows of U (SubspaceMin.h:256-267, the two
 * lbfgsx_b_wcombine calls) and the sweep's statements on those rows, walking the index list;
 * sums = {#L, #U, #P, 0, 0, #L with lambda < 0, #U with mu < 0} over those rows */
int lbfgsx_b_lu_sweep(lbfgsx_ctx* c, const double* coef, double theta, int64_t sums[7]);
/* element-wise statements of SubspaceMin.h selected by LBFGSX_SO_* */
int lbfgsx_b_sub_op(lbfgsx_ctx* c, int op);
/* copy the per-coordinate state byte (LBFGSX_ST_* bits) to the host: n bytes */
int lbfgsx_b_download_state(lbfgsx_ctx* c, unsigned char* host);
/* drt.dot(g) (SubspaceMin.h:281,289) */
int lbfgsx_b_dot_drt_g(lbfgsx_ctx* c, double* dg);
/* drt = xcp - x [, drt.normalize()]  (LBFGSB.h:163-164,191) */
int lbfgsx_b_dir_from_xcp(lbfgsx_ctx* c, int normalize);

/* ---- lock-step batch (BASELINE.json cfg5) ------------------------------------------------------------
 * P independent problems of equal dimension advance together: one kernel launch per reference statement for the
 * whole batch, grid = (chunks, problems), per-problem scalars / reduction workspace / buffer roles / history ring.
 * The host (include/LBFGSBatched.h) keeps one line-search state machine per problem and fills one descriptor per
 * problem and launch.  No reference counterpart (the reference solves one problem per call); per problem the
 * arithmetic is that of LBFGS.h:78-173 with LineSearchMoreThuente. */
typedef struct lbfgsx_batch lbfgsx_batch;
typedef struct
{
    int active;   /* 0: this problem sits out of the launch */
    int mode;     /* two-loop step kind: 0 INIT q=a*g, 1 SUB q-=alpha*y, 2 SUBDIV (q-alpha*y)/theta, 3 ADD q+=(alpha-beta)*s */
    int x_in;     /* point index (0..2): xp for a trial / post, the current point for eval / two-loop */
    int x_out;    /* point index written by a trial; the accepted point for post */
    int col_u;    /* physical history column of the update vector (post: the spare column to fill) */
    int col_w;    /* physical history column of the dot product, -1 = gradient at x_in */
    int i_num, i_den, i_num2, i_theta, i_out; /* indices into the problem's scalar table */
    float pad;
    double step;  /* trial step, or the scale a of the two-loop INIT */
} lbfgsx_bat_desc;
enum
{
    LBFGSX_BAT_EVAL = 0, LBFGSX_BAT_TRIAL = 1, LBFGSX_BAT_POST = 2, LBFGSX_BAT_TWOLOOP = 3,
    /* a user objective evaluated by the caller over the whole batch: the trial point alone (x_out = x_in + step * drt),
     * then -- after the caller's kernels have written f and grad of every active problem -- grad(x_out) . drt, or, after
     * the first evaluation, grad . grad and x . x at x_in */
    LBFGSX_BAT_POINT = 4, LBFGSX_BAT_GDOT = 5, LBFGSX_BAT_NORMS = 6
};
/* per-problem description of a whole apply_Hv (BFGSMat.h:276-302): d = -H grad(x_in), with the history columns
 * pcol[0..ncorr) newest -> oldest (physical column ids) */
typedef struct
{
    int active;
    int x_in;   /* which of the 3 points holds the gradient */
    int ncorr;
    int pcol[32];
} lbfgsx_bat_hvdesc;
/* per-problem description of a whole lock-step iteration (lbfgsx_bat_iterate) */
enum
{
    LBFGSX_BAT_IT_POST = 1,      /* the statements after a finished line search first (xp -> cur, pair into column `spare`) */
    LBFGSX_BAT_IT_POST_ONLY = 2, /* ... and nothing else: the caller already knows that this problem stops */
    LBFGSX_BAT_IT_TRIAL = 4,     /* the first trial of the next line search last (built-in objectives only) */
    LBFGSX_BAT_IT_TRIAL_ONLY = 8 /* nothing but a further trial of a search that goes on: x_trial = x(xp) + step * drt (drt as
                                  * the launch that opened the search left it), results 5 and 6.  Lets problems at different
                                  * points of their iteration share the launch */
};
#define LBFGSX_BAT_NRES 8        /* doubles per problem in the result table of a launch */
typedef struct
{
    int active;
    int flags;    /* LBFGSX_BAT_IT_* */
    int cur;      /* point (0..2) holding the accepted iterate and its gradient */
    int xp;       /* POST: the point the finished line search started from */
    int trial;    /* TRIAL: the point that receives cur + step * drt and the gradient there */
    int ncorr;    /* pairs stored BEFORE this launch */
    int spare;    /* POST: physical column that receives (s, y) */
    int pad;
    double step;  /* TRIAL: the step */
    int pcol[32]; /* physical columns of the stored pairs, newest -> oldest */
} lbfgsx_bat_itdesc;
/* history lengths: the reference puts no upper bound on m (Param.h:350-376); here
 *   LBFGSX_MAX_M_BOUNDED  an L-BFGS-B context passes the 2c coefficients of a W product in kernel arguments (80 slots) and keeps
 *                         the host's 2c-vectors in arrays of that size: lbfgsx_create(LBFGSX_FLAG_BOUNDED) refuses m > 40
 *                         (tests/test_lbfgsb_gpu.py: trajectories at m = 40, the refusal at 41);
 *   LBFGSX_MAX_M_BATCH    the lock-step batch's descriptors carry 32 column ids: lbfgsx_bat_create refuses m > 31;
 * unconstrained L-BFGS contexts take any m (the persistent one-launch recursion describes up to 128 pairs, beyond that the
 * step launches run). */
#define LBFGSX_MAX_M_BOUNDED 40
#define LBFGSX_MAX_M_BATCH 31
int lbfgsx_bat_create(lbfgsx_batch** out, int dtype, int64_t n, int m, int nproblems, int device);
void lbfgsx_bat_destroy(lbfgsx_batch* c);
/* a created batch made ready for another set of problems of the same shape (scalars and counters cleared, nothing
 * re-allocated): what lets a caller keep the ~(2m+9) n P elements of one batch alive across minimisations */
int lbfgsx_bat_reset(lbfgsx_batch* c);
/* index of a scalar inside a problem's table: kind 0 = ys[col], 1 = theta[col], 2 = two-loop dot k, 3 = output k */
int lbfgsx_bat_scalar_index(const lbfgsx_batch* c, int kind, int k);
/* x0 of problem p (point 0) = extended-Rosenbrock start for seed seed0 + p */
int lbfgsx_bat_gen_rosen_x0(lbfgsx_batch* c, uint64_t seed0);
/* the diagonal quadratics of seeds seed0 + p (a, b as lbfgsx_gen_diag_quad makes them for one problem) and x0 = 0 */
int lbfgsx_bat_gen_diag_quad(lbfgsx_batch* c, double kappa, uint64_t seed0);
/* device pointers into the batch for code that evaluates its own objective: kind 0 = x, 1 = grad of `point` (0..2) of
 * `problem`, 2 = its search direction; consecutive problems of one point lie lbfgsx_bat_ld() elements apart.  The
 * caller's kernels must be ordered after the library's on lbfgsx_bat_stream() (launch there, or synchronise). */
void* lbfgsx_bat_vec(lbfgsx_batch* c, int kind, int point, int problem);
int64_t lbfgsx_bat_ld(const lbfgsx_batch* c);
void* lbfgsx_bat_stream(lbfgsx_batch* c);
/* one launch for the whole batch; desc = P descriptors; then out[p*nout + k] = scalar (desc[p].i_out + k) */
int lbfgsx_bat_launch(lbfgsx_batch* c, int kind, int objective, const lbfgsx_bat_desc* desc, int nout, double* out);
/* The whole two-loop recursion of every active problem in ONE launch: one 256-thread block per problem keeps its q
 * vector in registers across the 2c+1 steps (block-level reductions only, no grid synchronisation), so the history
 * is read once and q never travels: (4c+2) n elements instead of (8c+1) n.  Same element-wise arithmetic and
 * order-independent reductions as the step kernels, hence bit-identical results.  The final dot (grad . d) lands in
 * scalar dot(2*ncorr) of each problem, as after the step-wise sequence.  Applicable when the vector fits the block's
 * registers (n a multiple of the 16-byte vector width and n <= 256 * 98 * width: 100352 floats / 50176 doubles);
 * LBFGSX_E_INVALID otherwise -- the caller then issues the LBFGSX_BAT_TWOLOOP steps. */
int lbfgsx_bat_apply_Hv(lbfgsx_batch* c, const lbfgsx_bat_hvdesc* desc);
/* ONE launch per lock-step iteration: for every active problem, one 256-thread block runs
 *   [POST]   s = x - xp, y = grad - gradp into the spare column; grad.grad, x.x, s.y, y.y      (LBFGS.h:130,137,159-161)
 *            and decides add_correction's test s.y > eps * y.y itself (LBFGS.h:161, BFGSMat.h:83-97)
 *   always   drt = -H grad by the two-loop recursion over the history that results (BFGSMat.h:276-302), the direction
 *            held in registers / LDS from the first statement to the last; grad . drt
 *   [TRIAL]  x_trial = x + step * drt, f and grad there, grad_trial . drt   (LineSearchMoreThuente.h:412-414,
 *            LineSearchNocedalWright.h:146-148) -- the first trial of the next line search, whose step the caller knows
 * with the element-wise arithmetic and the order-independent sums of the statement-wise launches, hence the same bits.
 * out[p * LBFGSX_BAT_NRES + k], k = 0..6: grad.grad, x.x, s.y, y.y (POST), grad.drt, f_trial, grad_trial.drt (TRIAL).
 * The caller's host logic stays the reference's: it reads the sums, applies the stopping tests (a problem that stops has
 * had its direction and trial computed in vain, nothing else), rotates the ring when s.y > eps * y.y, and feeds the trial to the
 * line search it starts.  A problem longer than one block's registers hold (100 352 floats / 50 176 doubles) is split over
 * 2..16 consecutive blocks, each with its share of the direction resident, and the blocks exchange their partial sums at every
 * step (n up to 1 605 632 floats / 802 816 doubles, a multiple of the 16-byte vector width; m <= 31): lbfgsx_bat_iterate_ok.
 * LBFGSX_E_INVALID otherwise -- the caller then issues the statement-wise launches.  LBFGSX_E_RUNTIME: the blocks of a split
 * problem were not resident together within 100 ms (something else held the CUs); no sums are returned -- a part that gives up
 * stops publishing and sets the launch's error word, its siblings give up on seeing it, the reporting part checks the word
 * (environment LBFGSX_BAT_DEBUG_XCH_FAULT=k at lbfgsx_bat_create: the k-th such launch of the batch behaves like that). */
int lbfgsx_bat_iterate(lbfgsx_batch* c, int objective, const lbfgsx_bat_itdesc* desc, double* out);
int lbfgsx_bat_iterate_ok(const lbfgsx_batch* c);
/* instrumentation: enable != 0 brackets every launch of the batch with a pair of events; lbfgsx_bat_timing_read waits for
 * the stream and returns {sum of the launches' durations in ms, launches, host waits, waits that timed out} since the
 * last read */
int lbfgsx_bat_timing(lbfgsx_batch* c, int enable);
int lbfgsx_bat_timing_read(lbfgsx_batch* c, double out[4]);
/* out[p] = scalar idx[p] of problem p */
int lbfgsx_bat_fetch(lbfgsx_batch* c, const int* idx, double* out);
int lbfgsx_bat_download_x(lbfgsx_batch* c, int p, int point, void* host);
int lbfgsx_bat_sync(lbfgsx_batch* c);

/* ---- the exchange step of the batched mode over RCCL --------------------------------------
 * After a batch has been sharded over the GPUs of a node (lbfgsx_batch_minimize_lockstep_multi in lbfgsx_solver.h, or one
 * process per GPU) the only data that crosses devices are the per-problem result records.  This call leaves ALL `count`
 * records (rec_bytes each, in problem-id order, given in host memory as the solver returned them) on EVERY listed device:
 * device r contributes its contiguous block, one grouped ncclAllGather over xGMI moves the blocks, dev_out[r] receives a
 * device buffer on devices[r] (release it with lbfgsx_device_free).  One rank per GPU: a device listed twice is refused.
 * RCCL is loaded with dlopen on first use.  No reference counterpart (the reference solves one problem per call). */
int lbfgsx_rccl_allgather_records(const int* devices, int ndev, const void* records, int64_t count, int64_t rec_bytes,
                                  void** dev_out);
/* ---- the sum over the row shards of ONE problem, natively over RCCL --------------------------------
 * A problem whose rows are spread over several GPUs (lbfgsx_set_shard) needs every n-length sum of the driver -- the
 * reference's fx, grad.dot(drt), grad.norm(), x.norm(), s.y, y.y (LBFGS.h:92,123,130,161) and the 6m + 7 sums of the
 * Gram-space pass -- added over the shards before any scalar logic runs.  lbfgsx_comm_allreduce_sum does that for one
 * rank's small bundle of doubles (<= 512): device buffer, ONE ncclAllReduce(sum, f64) over xGMI on the rank's stream,
 * result back in `buf`; every rank gets the same bits.  Communicators:
 *   lbfgsx_comm_create_local  all ranks in THIS process, one host thread per device (ncclCommInitAll).  A device listed
 *                             twice cannot be two RCCL ranks: the sums are then formed in host memory, in rank order,
 *                             behind a barrier (the emulation a one-GPU box runs; lbfgsx_comm_info tells which);
 *   lbfgsx_comm_create_rank   one rank of a multi-process communicator (ncclCommInitRank; the id comes from
 *                             lbfgsx_comm_unique_id on one rank and travels through the caller's launcher).
 * Every rank must make the same sequence of calls (as with any collective).  lbfgsx_comm_abort releases ranks that wait
 * for one that failed.  RCCL is loaded with dlopen on first use.  No reference counterpart. */
typedef struct lbfgsx_comm lbfgsx_comm;
int lbfgsx_comm_unique_id(unsigned char id[128]);
int lbfgsx_comm_create_rank(lbfgsx_comm** out, int device, int rank, int nranks, const unsigned char id[128]);
int lbfgsx_comm_create_local(lbfgsx_comm** out, const int* devices, int ndev);
int lbfgsx_comm_allreduce_sum(lbfgsx_comm* comm, int local_rank, double* buf, int count);
int lbfgsx_comm_abort(lbfgsx_comm* comm);
/* the same, remembering WHICH local rank failed first (the one whose error is the root cause; the others then fail with
 * "aborted by another rank"): lbfgsx_comm_first_abort returns it, -1 when nobody called lbfgsx_comm_abort_from.  A rank of
 * another PROCESS that dies cannot call either: the surviving ranks leave their wait through RCCL's asynchronous error or
 * after LBFGSX_COMM_TIMEOUT_S seconds (default 300) and abort their own communicator. */
int lbfgsx_comm_abort_from(lbfgsx_comm* comm, int local_rank);
int lbfgsx_comm_first_abort(const lbfgsx_comm* comm);
/* info = {ranks, ranks driven by this process, 1 when RCCL carries the sums (0: host emulation), RCCL version code} */
int lbfgsx_comm_info(const lbfgsx_comm* comm, int info[4]);
int64_t lbfgsx_comm_calls(const lbfgsx_comm* comm, int local_rank);
/* the all-reduce as the callback LBFGSSolver::set_reducer / lbfgsx_solver_set_allreduce take: pass
 * lbfgsx_comm_allreduce_hook with user = lbfgsx_comm_hook_arg(comm, local_rank) (owned by the communicator) */
void* lbfgsx_comm_hook_arg(lbfgsx_comm* comm, int local_rank);
void lbfgsx_comm_allreduce_hook(double* buf, int count, void* hook_arg);
void lbfgsx_comm_destroy(lbfgsx_comm* comm);
int lbfgsx_device_download(int device, const void* dev_ptr, int64_t bytes, void* host);
void lbfgsx_device_free(int device, void* dev_ptr);

/* ---- instrumentation ----------------------------------------------------------------------------------*/
/* average duration (ms) of the two-loop step kernels since the last reset, measured with HIP events on
 * the context's stream; count = number of timed launches.  on = 2: one event pair per apply_Hv only (the step launches
 * then run back to back, without an event between them) and the per-step figures are that time / (2c+1) */
int lbfgsx_timing_enable(lbfgsx_ctx* c, int on);
int lbfgsx_timing_read(lbfgsx_ctx* c, double* twoloop_ms_total, int64_t* twoloop_launches,
                       double* applyhv_ms_total, int64_t* applyhv_calls);
/* number of apply_Hv calls of this context served by the persistent one-launch kernel (k_twoloop_persist: the
 * whole recursion in one launch of occupancy x CUs blocks with part of q resident on the CUs).  It is used when
 * m <= 128 and LBFGSX_PERSIST != 0, one such kernel per device at a time within the process (a per-device lock taken
 * for the launch; a context that finds it taken issues its 2c+1 step launches for that call, as does one whose launch
 * timed out because another process shares the GPU).  Results are bit-identical either way. */
int64_t lbfgsx_persistent_launches(const lbfgsx_ctx* c);
/* of the apply_Hv calls timed since lbfgsx_timing_enable, how many were persistent launches that also carried the post
 * statements (lbfgsx_post_linesearch_spec): their algorithmic bytes are (8c+1) n for the product plus 6n - 2n for K3 */
int64_t lbfgsx_timing_fused_launches(const lbfgsx_ctx* c);
/* elements of q (= the direction vector, BFGSMat.h:283-301 `res`) that a persistent launch of this context keeps in the
 * registers / LDS of the CUs for the whole recursion: that share of q's traffic never reaches HBM (0: not available) */
int64_t lbfgsx_persistent_resident_elems(const lbfgsx_ctx* c);
/* Diagnostic: runs the grid-wide reduction every kernel of the path ends in (csrc/reduce.cuh) on `nred` sums of known small
 * integers over `grid` blocks of 256 threads; out[r] = the total of sum r, which the caller checks against the closed
 * form: sum over g < 256 grid of (r + 1)(1 + g mod 7) + (block(g) mod 3) + (1000003 r mod 17).  nred in {1 2 3 5 7 8 9 25 31
 * 33 40 50 56}; f32_accumulators != 0 uses the accumulator type of f32 contexts. */
int lbfgsx_selftest_reduce(lbfgsx_ctx* c, int nred, int grid, int f32_accumulators, double* out);
/* STREAM-style device bandwidth probe on this context's vectors: copy (XT = X) and triad, GB/s */
int lbfgsx_stream_probe(lbfgsx_ctx* c, int reps, double* copy_gbs, double* triad_gbs);

#ifdef __cplusplus
}
#endif
#endif /* LBFGSX_H */
