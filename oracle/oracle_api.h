/* oracle/oracle_api.h -- TEST INFRASTRUCTURE ONLY (C ABI shared by oracle/_ref and the restatement).
 *
 * Two implementations export this same surface with different prefixes:
 *   oracle_ref_*   built by oracle/Makefile from the UNMODIFIED reference headers under
 *                  /root/reference/include + oracle/eigen_shim  ->  oracle/_ref/libref_<acc>.so
 *   oracle_port_*  the self-contained restatement oracle/lbfgs_oracle.cpp -> oracle/liboracle_<acc>.so
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load either.
 */
#ifndef LBFGSX_ORACLE_API_H
#define LBFGSX_ORACLE_API_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_F64 = 0, ORACLE_F32 = 1 };
enum { ORACLE_LS_NOCEDAL_WRIGHT = 0, ORACLE_LS_MORE_THUENTE = 1, ORACLE_LS_BACKTRACKING = 2, ORACLE_LS_BRACKETING = 3 };
enum { ORACLE_OBJ_DIAG_QUAD = 0, ORACLE_OBJ_EXT_ROSENBROCK = 1 };

/* mirrors LBFGSParam / LBFGSBParam (reference include/LBFGSpp/Param.h:168-184, 327-343) */
typedef struct
{
    int m;
    double epsilon, epsilon_rel;
    int past;
    double delta;
    int max_iterations;
    int linesearch; /* LINE_SEARCH_TERMINATION_CONDITION, L-BFGS only */
    int max_linesearch;
    double min_step, max_step, ftol, wolfe;
    int max_submin; /* L-BFGS-B only */
} oracle_params;

typedef struct
{
    int niter;   /* return value of minimize() */
    int nfev;    /* number of functor calls */
    double fx;   /* final objective */
    double gnorm; /* final_grad_norm() */
    int status;  /* 0 ok, 1 std::invalid_argument, 2 std::logic_error, 3 std::runtime_error, 4 other */
    char msg[200];
} oracle_result;

/* per-evaluation trace: fx[k] and x[0::stride] of the k-th functor call (k < cap) */
typedef struct
{
    int cap;
    int count;
    double* fx;  /* [cap] */
    long stride; /* >=1 */
    long nsamp;  /* ceil(n/stride) */
    double* xs;  /* [cap*nsamp], may be NULL */
} oracle_trace;

#define ORACLE_DECL(prefix)                                                                                       \
    int prefix##_lbfgs(int dtype, int ls, int obj, long n, const void* a, const void* b, void* x,                 \
                       const oracle_params* p, oracle_trace* tr, oracle_result* out);                             \
    int prefix##_lbfgsb(int dtype, int obj, long n, const void* a, const void* b, const void* lb, const void* ub, \
                        void* x, const oracle_params* p, oracle_trace* tr, oracle_result* out);                   \
    /* feed npairs (s,y) columns through add_correction, then res = alpha*H*v (BFGSMat.h:81-97,276-302) */       \
    int prefix##_apply_Hv(int dtype, long n, int m, int npairs, const void* S, const void* Y, const void* v,      \
                          double alpha, void* res);                                                               \
    /* objective only: returns fx, writes grad */                                                                \
    double prefix##_eval(int dtype, int obj, long n, const void* a, const void* b, const void* x, void* grad);    \
    /* generalized Cauchy point + subspace minimisation on a history of npairs corrections                       \
       (Cauchy.h:86-284, SubspaceMin.h:122-302).  sets are returned as int arrays of capacity n. */              \
    int prefix##_cauchy_subspace(int dtype, long n, int m, int npairs, const void* S, const void* Y,              \
                                 const void* x0, const void* g, const void* lb, const void* ub, int max_submin,   \
                                 void* xcp, void* vecc, int* newact, int* n_newact, int* fv, int* n_fv,           \
                                 void* drt);                                                                      \
    /* minimize() followed by final_approx_hessian() / final_approx_inverse_hessian() (LBFGS.h:192-197),           \
       B and H column-major n x n doubles; -1000 when the implementation has no dense getters */                    \
    int prefix##_lbfgs_hessians(int dtype, int ls, int obj, long n, const void* a, const void* b, void* x,        \
                                const oracle_params* p, double* B, double* H, oracle_result* out);                \
    /* timing aid of bench.py's CPU baseline: stamps[k] = steady-clock seconds right after the k-th functor call of the   \
       following solves (k < cap); NULL switches it off */                                                             \
    int prefix##_set_eval_clock(double* stamps, int cap);                                                         \
    const char* prefix##_describe(void);

ORACLE_DECL(oracle_ref)
ORACLE_DECL(oracle_port)
/* restatement only: every n-length sum of the L-BFGS path is multiplied by r (a power of two), which makes the run
 * the exact image of the r-fold replicated problem -- see oracle/acc.h; 1 = off */
int oracle_port_set_replication(double r);
/* liboracle_native_omp.so (bench.py's all-cores timing baseline): set / query the OpenMP thread count; 0 elsewhere */
int oracle_port_set_threads(int nthreads);

#ifdef __cplusplus
}
#endif
#endif
