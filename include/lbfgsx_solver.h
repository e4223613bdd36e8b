/* include/lbfgsx_solver.h -- C ABI of liblbfgsx_solver.so: the drop-in C++ solvers (include/LBFGS.h,
 * include/LBFGSB.h) instantiated for the built-in device objectives, for callers without a C++ compiler
 * (ctypes / cgo / JNI ...).  Mirrors LBFGSSolver<T, LineSearch>::minimize (reference LBFGS.h:78-173) and
 * LBFGSBSolver<T>::minimize (reference LBFGSB.h:116-262): same parameters (LBFGSParam / LBFGSBParam fields),
 * same return value (iteration count), exceptions mapped to status codes + message.
 */
#ifndef LBFGSX_SOLVER_H
#define LBFGSX_SOLVER_H

#include <stdint.h>

#include "lbfgsx.h"

#ifdef __cplusplus
extern "C" {
#endif

enum
{
    LBFGSX_LS_NOCEDAL_WRIGHT = 0,
    LBFGSX_LS_MORE_THUENTE = 1,
    LBFGSX_LS_BACKTRACKING = 2,
    LBFGSX_LS_BRACKETING = 3
};
enum { LBFGSX_ALGO_LBFGS = 0, LBFGSX_ALGO_LBFGSB = 1 };

/* LBFGSParam / LBFGSBParam fields (reference Param.h:79-161, 236-320); doubles are cast to the scalar type */
typedef struct
{
    int m;
    double epsilon, epsilon_rel;
    int past;
    double delta;
    int max_iterations;
    int linesearch;
    int max_linesearch;
    double min_step, max_step, ftol, wolfe;
    int max_submin;
} lbfgsx_params;

typedef struct
{
    int niter;    /* minimize() return value */
    int nfev;     /* objective evaluations */
    double fx;    /* final objective value */
    double gnorm; /* final_grad_norm() */
    int status;   /* 0, or LBFGSX_E_* of the exception the reference API would have thrown */
    char msg[200];
} lbfgsx_result;

/* optional per-evaluation trace (parity testing): fx[k] and x[0::stride] at the k-th objective evaluation */
typedef struct
{
    int cap;
    int count;
    double* fx;
    int64_t stride;
    int64_t nsamp;
    double* xs;
} lbfgsx_trace;

typedef struct lbfgsx_solver lbfgsx_solver;

/* constructor of LBFGSSolver / LBFGSBSolver: validates the parameters (check_param) */
int lbfgsx_solver_create(lbfgsx_solver** out, int algo, int dtype, int linesearch, const lbfgsx_params* p, int device);
const char* lbfgsx_solver_create_error(void); /* message of a failed lbfgsx_solver_create */
void lbfgsx_solver_destroy(lbfgsx_solver* s);
/* allocate device state for dimension n and expose it (to generate / upload resident inputs) */
int lbfgsx_solver_prepare(lbfgsx_solver* s, int64_t n);
lbfgsx_ctx* lbfgsx_solver_ctx(lbfgsx_solver* s);
/* progress hook: fn(k, user) runs on the host after iteration k has produced the next search direction */
int lbfgsx_solver_set_iteration_hook(lbfgsx_solver* s, void (*fn)(int, void*), void* user);
/* LBFGSSolver::set_recursion (extension): 0 = vector two-loop (bit-parity path, default), 1 = Gram-space form
 * (include/LBFGSpp/GramSpace.h), 2 = Gram-space form with an f32 history (f64 solvers only).  LBFGSX_E_INVALID for
 * L-BFGS-B solvers. */
int lbfgsx_solver_set_recursion(lbfgsx_solver* s, int form);
/* LBFGSSolver::set_reducer (extension, row-sharded runs): fn(values, count, user) must sum `values` over all ranks in
 * place (an all-reduce); NULL switches back.  L-BFGS solvers with the Gram-space recursion only. */
int lbfgsx_solver_set_allreduce(lbfgsx_solver* s, void (*fn)(double*, int, void*), void* user);
/* LBFGSSolver::set_devices (extension): the next lbfgsx_solver_minimize with a host x row-shards the ONE problem over the
 * listed GPUs of this node -- one host thread + context per device, the driver's sums through lbfgsx_comm_allreduce_sum
 * (RCCL; include/lbfgsx.h).  ndev = 0 switches back.  L-BFGS solvers only. */
int lbfgsx_solver_set_devices(lbfgsx_solver* s, const int* devices, int ndev);
/* test entry: Cauchy::get_cauchy_point + SubspaceMin::subspace_minimize of the drop-in headers on a history of
 * npairs host-provided corrections; counts = {|newact|, |free|, crossings, BOXCQP sweeps} */
int lbfgsx_test_cauchy_subspace(int dtype, int64_t n, int m, int npairs, const void* S, const void* Y, const void* x0,
                                const void* g, const void* lb, const void* ub, int max_submin, void* xcp, double* vecc,
                                unsigned char* state, void* drt, long long counts[4], char* errbuf, int errlen);
/* final_approx_hessian() / final_approx_inverse_hessian() of the last L-BFGS minimize() (reference LBFGS.h:192-197):
 * column-major n x n doubles; small n only */
int lbfgsx_solver_hessians(lbfgsx_solver* s, double* B, double* H);
/* L-BFGS-B instrumentation of the last minimize(): {GCP break points crossed, BOXCQP sweeps, subspace calls,
 * unconverged subspace calls, BFGS resets, 0, 0, 0} */
int lbfgsx_solver_stats(lbfgsx_solver* s, long long out[8]);
/* more of the same: {GCP crossings handled by the device search, partial sorts that had to be redone in full,
 * searches served by a partial sort, subspace us, line-search us, add_correction us, 0, 0} */
int lbfgsx_solver_stats2(lbfgsx_solver* s, long long out[8]);
/* and: {GCP searches, sum of their finite positive break points (the reference's |ord|, Cauchy.h:132-133), sum of the
 * break points actually sorted, 0, 0, 0, 0, 0} */
int lbfgsx_solver_stats3(lbfgsx_solver* s, long long out[8]);
/* minimize(): objective = LBFGSX_OBJ_*; a/b host arrays or NULL (resident); x host in/out or NULL (resident:
 * start point in LBFGSX_VEC_X, result left there); lb/ub host arrays or NULL (resident), L-BFGS-B only */
int lbfgsx_solver_minimize(lbfgsx_solver* s, int objective, int64_t n, const void* a, const void* b, void* x,
                           const void* lb, const void* ub, lbfgsx_trace* trace, lbfgsx_result* out);

/* ---- batched mode (BASELINE.json cfg5): many independent minimisations on one GPU ------------------------
 * Problem `id` is the extended Rosenbrock (or diag quadratic) instance generated on the device from seed
 * `seed_base + id`.  `nthreads` host workers each own one solver/context/stream and pull
 * problem ids from a shared counter, so up to `nthreads` problems are resident and overlap on the GPU; there is
 * no reference counterpart (the reference solves one problem per call).  One process per GPU shards a larger
 * batch by giving each rank its own [first, first+count) range: no inter-GPU traffic on the critical path. */
typedef struct
{
    int niter, nfev, status;
    double fx, gnorm;
} lbfgsx_batch_item;
int lbfgsx_batch_minimize(int algo, int dtype, int linesearch, const lbfgsx_params* p, int objective, int64_t n,
                          int64_t first, int64_t count, uint64_t seed_base, int device, int nthreads,
                          lbfgsx_batch_item* out);
/* lock-step variant (include/LBFGSBatched.h): all `count` problems resident at once and advanced together, one
 * kernel launch per statement for the whole batch; L-BFGS + LineSearchMoreThuente + extended Rosenbrock (the general
 * form below takes the line search and the objective).
 * x_out (optional): count*n scalars receiving the final iterates. */
int lbfgsx_batch_minimize_lockstep(int dtype, const lbfgsx_params* p, int64_t n, int64_t first, int count,
                                   uint64_t seed_base, int device, lbfgsx_batch_item* out, void* x_out, char* errbuf,
                                   int errlen);
/* The same over several GPUs of one node from one process (BASELINE.json cfg5 "sharded over 8 x MI355X"):
 * devices[r] (r < ndev) solves the r-th contiguous, balanced block of the `count` problem ids in its own lock-step
 * batch on its own host thread; no data crosses between devices, the records (and x_out) are gathered in host memory in
 * problem-id order.  A device id may repeat.  Records are bit-identical to the single-device call. */
int lbfgsx_batch_minimize_lockstep_multi(int dtype, const lbfgsx_params* p, int64_t n, int64_t first, int count,
                                         uint64_t seed_base, const int* devices, int ndev, lbfgsx_batch_item* out,
                                         void* x_out, char* errbuf, int errlen);
/* The general form: the line search the reference's solver takes as its template parameter (LBFGS.h:20-21) -- the two
 * policies that exist as state machines, LBFGSX_LS_MORE_THUENTE and LBFGSX_LS_NOCEDAL_WRIGHT -- and the built-in
 * objective: LBFGSX_OBJ_EXT_ROSENBROCK (start points of seed seed_base + id) or LBFGSX_OBJ_DIAG_QUAD (a, b of
 * lbfgsx_gen_diag_quad(kappa, seed_base + id), x0 = 0).  Every problem follows the trajectory of the single-problem solver
 * with that policy, bit for bit.  (A user objective on device memory: LBFGSBatchedSolver::minimize(BatchFunctor, ...) in
 * include/LBFGSBatched.h, over lbfgsx_bat_launch(LBFGSX_BAT_POINT / _GDOT / _NORMS).) */
int lbfgsx_batch_minimize_lockstep_ex(int dtype, int linesearch, int objective, double kappa, const lbfgsx_params* p, int64_t n,
                                      int64_t first, int count, uint64_t seed_base, const int* devices, int ndev,
                                      lbfgsx_batch_item* out, void* x_out, char* errbuf, int errlen);

/* A lock-step batch kept alive across minimisations: `count` problems of dimension n resident on `device` (about
 * (2m + 9) n count scalars of HBM, allocated here once).  Every lbfgsx_lockstep_minimize solves the problems of ids
 * [first, first + count) of seed_base from their start points again -- what a service that solves batch after batch of the
 * same shape calls, and what bench.py times (the allocation of ~12 GB is setup, not solve).  timing != 0: events around
 * every launch; stats = {lock-step iterations, 1 if the one-launch-per-iteration form ran, sum of the launches'
 * durations in ms (timing), launches, host waits, waits that timed out, 0, 0}.  No reference counterpart. */
typedef struct lbfgsx_lockstep lbfgsx_lockstep;
int lbfgsx_lockstep_create(lbfgsx_lockstep** out, int dtype, int linesearch, const lbfgsx_params* p, int64_t n, int count,
                           int device, int timing, char* errbuf, int errlen);
int lbfgsx_lockstep_minimize(lbfgsx_lockstep* h, int objective, double kappa, uint64_t seed_base, int64_t first,
                             lbfgsx_batch_item* out, void* x_out, double stats[8], char* errbuf, int errlen);
int lbfgsx_lockstep_set_timing(lbfgsx_lockstep* h, int on);  /* for the minimisations that follow */
void lbfgsx_lockstep_destroy(lbfgsx_lockstep* h);

#ifdef __cplusplus
}
#endif
#endif
