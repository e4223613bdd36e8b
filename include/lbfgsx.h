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

/* ---- instrumentation ----------------------------------------------------------------------------------*/
/* average duration (ms) of the two-loop step kernels since the last reset, measured with HIP events on
 * the context's stream; count = number of timed launches */
int lbfgsx_timing_enable(lbfgsx_ctx* c, int on);
int lbfgsx_timing_read(lbfgsx_ctx* c, double* twoloop_ms_total, int64_t* twoloop_launches,
                       double* applyhv_ms_total, int64_t* applyhv_calls);
/* STREAM-style device bandwidth probe on this context's vectors: copy (XT = X) and triad, GB/s */
int lbfgsx_stream_probe(lbfgsx_ctx* c, int reps, double* copy_gbs, double* triad_gbs);

#ifdef __cplusplus
}
#endif
#endif /* LBFGSX_H */
