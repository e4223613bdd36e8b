// lbfgspp_amd/csrc/lbfgsb_x.hip -- instantiations and launchers of the m-generic L-BFGS-B passes (lbfgsb_x.cuh).
#include <algorithm>
#include <cstdlib>

#include "lbfgsb_x.hpp"

namespace lbfgsx {
namespace xl {

// column classes: (NCL columns per lane, G lanes per row) for 2c <= NCL * G
#ifndef LBFGSX_X40
#define LBFGSX_X40(CALL) CALL(10, 4)  // 2c in 33..40 (m = 17..20); -D'LBFGSX_X40(CALL)=CALL(20, 2)' builds the other split
#endif
#define LBFGSX_XCLASS(ncols, CALL)             \
    do                                         \
    {                                          \
        if ((ncols) <= 8) { CALL(4, 2); }      \
        else if ((ncols) <= 12) { CALL(6, 2); }  \
        else if ((ncols) <= 16) { CALL(8, 2); }  \
        else if ((ncols) <= 20) { CALL(10, 2); } \
        else if ((ncols) <= 24) { CALL(12, 2); } \
        else if ((ncols) <= 32) { CALL(8, 4); }  \
        else if ((ncols) <= 40) { LBFGSX_X40(CALL); } \
        else if ((ncols) <= 48) { CALL(12, 4); } \
        else if ((ncols) <= 60) { CALL(15, 4); } \
        else { CALL(20, 4); }                  \
    } while (0)

// The sweeps' solves at 2c = 33..40 split a row over two lanes of 20 columns: kx_solve_sweep<.., 0 ..> 472 -> 450 us, <.., 1 ..>
// 518 -> 469 us at m = 20 against four lanes of 10, while kx_rows goes the other way (438 -> 468 us) and the W'd pass is even
// (profiles/r6_cfg4_m20_ab.txt) -- so only the solves take the other split.  Any split gives the same bits (order-independent sums).
#ifndef LBFGSX_X40_SWEEP
#define LBFGSX_X40_SWEEP(CALL) CALL(20, 2)
#endif
#ifndef LBFGSX_X60_SWEEP
#define LBFGSX_X60_SWEEP(CALL) CALL(30, 2)  // 2c = 41..60 (m = 21..30)
#endif
#ifndef LBFGSX_X32_SWEEP
#define LBFGSX_X32_SWEEP(CALL) CALL(16, 2)  // 2c = 25..32 (m = 13..16): the same question, the same answer (profiles/r6_cfg4_m20_ab.txt)
#endif
#ifndef LBFGSX_X20_SWEEP
#define LBFGSX_X20_SWEEP(CALL) CALL(10, 2)
#endif
#define LBFGSX_XCLASS_SWEEP(ncols, CALL)                                  \
    do                                                                    \
    {                                                                     \
        if ((ncols) > 40 && (ncols) <= 48) { CALL(24, 2); }               \
        else if ((ncols) > 48 && (ncols) <= 60) { LBFGSX_X60_SWEEP(CALL); } \
        else if ((ncols) > 32 && (ncols) <= 40) { LBFGSX_X40_SWEEP(CALL); } \
        else if ((ncols) > 24 && (ncols) <= 32) { LBFGSX_X32_SWEEP(CALL); } \
        else if ((ncols) > 16 && (ncols) <= 20) { LBFGSX_X20_SWEEP(CALL); } \
        else LBFGSX_XCLASS(ncols, CALL);                                  \
    } while (0)

// blocks per CU: what the kernel was compiled for, but no more than two -- with the next trip's loads always in flight two
// blocks per CU saturate the memory system (a bare pass reads 6.27 TB/s at two, 6.1 at four / six), and every further
// block adds to the reduction's partials and to the bunching of the waves (scripts/experiments/kernels_x.hip: kx_rows 185 /
// 190 / 199 us at 2 / 3 / 4 blocks per CU)
// Blocks per CU, measured inside cfg4's iteration (the host's wait behind each pass, scripts/r5/waits_ab.sh,
// profiles/r5_grid_per_cu.txt): the passes that only stream -- by position over the compact copy -- are fastest at two blocks
// per CU (kx_solve_sweep<0>: 273 / 247 / 264 us at 1 / 2 / 3); the two that also GATHER vectors by row number -- the first sweep,
// which starts the compact vectors (293 / 312 / 319 us), and the W'd pass (260 / 273 / 282 us) -- are fastest at one: half as
// many gather streams in the air.  (scripts/experiments/kernels_x.hip, where everything is by position, says two for all.)
static inline int grid_rows(int64_t n, int rpw, int per_cu, int num_cus)
{
    if (const char* e = getenv("LBFGSX_X_PER_CU"))  // A/B
        per_cu = std::max(1, atoi(e));
    else
        per_cu = std::min(per_cu, 2);
    const int64_t per_block = int64_t(kWaves) * rpw;
    const int64_t want = (n + per_block - 1) / per_block;
    return int(std::max<int64_t>(1, std::min<int64_t>(want, std::min<int64_t>(int64_t(per_cu) * num_cus, kMaxGridX))));
}

template <class T>
int rows(hipStream_t s, int num_cus, int na, const ColsX<T>& cols, int ncols, const BVecs<T>& b, int vsel_id, int mask, int64_t n,
         const RedWsX& ws, double* out, double* out_dd, const ProX<T>& pro, const RowsX<T>& gr, int col_a, int col_b)
{
    if (ncols < 1 || ncols > kColsX || (na != 1 && na != 3))
        return LBFGSX_E_INVALID;
    {
        // byte model (ctx.hpp Counters::model_bytes): the columns over the rows walked; per row the state byte, the prologue's
        // vector and v (two vectors for the bound selectors), the position's row number where the columns are the compact
        // copy; written: rhs / cF where a prologue runs; the patching form reads the two fresh columns at the row and writes them
        // at the position
        const double e = sizeof(T);
        double per = double(ncols) * e + 1 + e;
        if (pro.mode != GP_NONE)
            per += 2 * e;                                   // pre read, rhs / cF written
        if (vsel_id == VS_LBOUND || vsel_id == VS_UBOUND)
            per += e;
        if (gr.in_idx)
            per += 4;
        if (na == 3 && gr.dst_a)
            per += 4 * e;
        model_add(per * double(n));
        if (gr.in_idx)
            model_compact_pass(n);
    }
#define CALL(NCL, G)                                                                                                           \
    if (na == 1 && gr.in_idx)                                                                                                  \
        LBFGSX_LAUNCH((kx_rows<T, NCL, G, 1, true>), dim3(grid_rows(n, 64 / G, occ_rows_x(NCL, G, 1), num_cus)), dim3(kBlock), 0, s, \
                      cols, ncols, b, vsel_id, mask, n, ws, out, out_dd, pro, gr, col_a, col_b);                               \
    else if (na == 1)                                                                                                          \
        LBFGSX_LAUNCH((kx_rows<T, NCL, G, 1, false>), dim3(grid_rows(n, 64 / G, occ_rows_x(NCL, G, 1), num_cus)), dim3(kBlock), 0, s, \
                      cols, ncols, b, vsel_id, mask, n, ws, out, out_dd, pro, gr, col_a, col_b);                               \
    else if (gr.in_idx && gr.dst_a)                                                                                            \
        LBFGSX_LAUNCH((kx_rows<T, NCL, G, 3, true, true>), dim3(grid_rows(n, 64 / G, occ_rows_x(NCL, G, 3, true), num_cus)), dim3(kBlock), 0, s, \
                      cols, ncols, b, vsel_id, mask, n, ws, out, out_dd, pro, gr, col_a, col_b);                               \
    else if (gr.in_idx)                                                                                                        \
        LBFGSX_LAUNCH((kx_rows<T, NCL, G, 3, true, false>), dim3(grid_rows(n, 64 / G, occ_rows_x(NCL, G, 3), num_cus)), dim3(kBlock), 0, s, \
                      cols, ncols, b, vsel_id, mask, n, ws, out, out_dd, pro, gr, col_a, col_b);                               \
    else if (gr.dst_a)                                                                                                         \
        LBFGSX_LAUNCH((kx_rows<T, NCL, G, 3, false, true>), dim3(grid_rows(n, 64 / G, occ_rows_x(NCL, G, 3, true), num_cus)), dim3(kBlock), 0, s, \
                      cols, ncols, b, vsel_id, mask, n, ws, out, out_dd, pro, gr, col_a, col_b);                               \
    else                                                                                                                       \
        LBFGSX_LAUNCH((kx_rows<T, NCL, G, 3, false, false>), dim3(grid_rows(n, 64 / G, occ_rows_x(NCL, G, 3), num_cus)), dim3(kBlock), 0, s, \
                      cols, ncols, b, vsel_id, mask, n, ws, out, out_dd, pro, gr, col_a, col_b)
    LBFGSX_XCLASS(ncols, CALL);
#undef CALL
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

template <class T>
int solve_sweep(hipStream_t s, int num_cus, int first, const ColsX<T>& cols, int ncols, const BVecs<T>& b, const BVecs<T>& bw,
                int vsel_id, const CoefX<T>& coef, int has_w, T theta, int64_t n, const RedWsX& ws, double* out, int* lu_list,
                unsigned* lu_cnt, unsigned lu_cap, const int* ridx, T* cli, T* cui, int cv, const ProX<T>* pro)
{
    if (ncols < 1 || ncols > kColsX)
        return LBFGSX_E_INVALID;
    const bool rhsk = pro && pro->mode == LBFGSX_GP_RHS && (pro->use1 || pro->use2);
    if (rhsk && (first || vsel_id != VS_NEG_RHS))
        return LBFGSX_E_INVALID;
    if (vsel_id == VS_LBOUND || vsel_id == VS_UBOUND || (cv == 2 && first) || (cv == 1 && !first))
        return LBFGSX_E_INVALID;  // v is one vector here; the compact vectors start with the first solve and are by position after it
    ProX<T> none;
    none.mode = LBFGSX_GP_NONE;
    none.use1 = none.use2 = 0;
    const ProX<T>& pr = rhsk ? *pro : none;  // (the arrays of `none` are never read)
    {
        // byte model: columns; read st, v's vector, cF, the two bounds (and x0 where they are not yet differences), y of the
        // sweep before; written y, rhs, st -- the first pass also yfallback and the two multipliers, and when it starts the
        // compact vectors (cv = 1) cF and the two bound differences of every position; vectors by row are gathers
        const double e = sizeof(T);
        double per = double(ncols) * e;
        const double rd = (1 + 4 * e) + (first ? 0 : e) + (cv == 2 ? 0 : e);
        const double wr = (1 + 2 * e) + (first ? 3 * e : 0) + (cv == 1 ? 3 * e : 0);
        per += wr + ((cv == 2 || !ridx) ? rd : 4.0);
        double tot = per * double(n);
        if (ridx && cv != 2)  // reads by row through the list of the positions' rows
            tot += (rd / e) * model_gather(n, n * 2, int(e)) + model_gather(n, n * 2, 1) - double(n);
        model_add(tot);
        if (ridx || cv)
            model_compact_pass(n);
    }
#define SWEEP(FIRST, IDX, RHSK, CVT)                                                                                              \
    LBFGSX_LAUNCH((kx_solve_sweep<T, NCL_, G_, FIRST, IDX, RHSK, CVT>),                                                            \
                  dim3(grid_rows(n, 64 / G_, FIRST ? 1 : occ_sweep_x(NCL_, G_, FIRST), num_cus)),                                      \
                  dim3(kBlock), 0, s, cols, ncols, b, bw, vsel_id, coef, has_w, theta, n, ws, out, lu_list, lu_cnt, lu_cap, ridx, cli,  \
                  cui, cv, pr)
#define CALL(NCL, G)                                                      \
    {                                                                     \
        constexpr int NCL_ = NCL, G_ = G;                                 \
        if (first && ridx) SWEEP(1, true, false, false);                  \
        else if (first) SWEEP(1, false, false, false);                    \
        else if (cv == 2)                                                 \
        {                                                                 \
            if (rhsk && ridx) SWEEP(0, true, true, true);                 \
            else if (rhsk) SWEEP(0, false, true, true);                   \
            else if (ridx) SWEEP(0, true, false, true);                   \
            else SWEEP(0, false, false, true);                            \
        }                                                                 \
        else if (rhsk && ridx) SWEEP(0, true, true, false);               \
        else if (rhsk) SWEEP(0, false, true, false);                      \
        else if (ridx) SWEEP(0, true, false, false);                      \
        else SWEEP(0, false, false, false);                               \
    }
    LBFGSX_XCLASS_SWEEP(ncols, CALL);
#undef CALL
#undef SWEEP
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

template <class T>
int multidot2_wf(hipStream_t s, int num_cus, const ColsX<T>& wfc, int ncols, int fresh_a, int fresh_b, const T* snew, const T* ynew,
                 const T* dvec, const int* idx, int64_t npos, const ColsX<T>& full, const int* list, int nlist, const RedWsX& ws,
                 double* out, T* dst_a, T* dst_b)
{
    if (ncols < 1 || ncols > kColsX)
        return LBFGSX_E_INVALID;
    // byte model: the compact columns over their positions + s_new, y_new, d gathered by the positions' rows + the row numbers;
    // the rows outside the copy: every column and two vectors at the row (one sector each)
    model_add(double(npos) * (double(ncols) * sizeof(T) + 4 + (dst_a ? 2.0 * sizeof(T) : 0.0)) +
              3.0 * model_gather(npos, npos * 2, int(sizeof(T))) + double(nlist) * 64.0 * (ncols + 2));
    model_compact_pass(npos);
    static const int dots_per_cu = [] { const char* e = getenv("LBFGSX_X_DOTS_PER_CU"); return e ? std::max(1, atoi(e)) : 1; }();  // see grid_rows
#define CALL(NCL, G)                                                                                                          \
    LBFGSX_LAUNCH((kx_multidot2_wf<T, NCL, G>),                                                                               \
                  dim3(grid_rows(std::max<int64_t>(npos, nlist), 64 / G, std::min(dots_per_cu, occ_dots_x(NCL)), num_cus)),   \
                  dim3(kBlock), 0, s, wfc, ncols, fresh_a, fresh_b, snew, ynew, dvec, idx, npos, full, list, nlist, ws, out, dst_a, dst_b)
    LBFGSX_XCLASS(ncols, CALL);
#undef CALL
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

template <class T>
int multidot2(hipStream_t s, int num_cus, const ColsX<T>& cols, int ncols, const T* v1, const T* v2, int64_t n, const RedWsX& ws,
              double* out)
{
    if (ncols < 1 || ncols > kColsX)
        return LBFGSX_E_INVALID;
    model_add(double(n) * (ncols + 2) * sizeof(T));  // byte model: full-length columns and the two vectors
#define CALL(NCL, G)                                                                                                        \
    LBFGSX_LAUNCH((kx_multidot2<T, NCL, G>), dim3(grid_rows(n, 64 / G, occ_dots_x(NCL), num_cus)), dim3(kBlock), 0, s, cols, ncols, \
                  v1, v2, n, ws, out)
    LBFGSX_XCLASS(ncols, CALL);
#undef CALL
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

template <class T>
int list2(hipStream_t s, int num_cus, const ColsX<T>& cols, int ncols, const BVecs<T>& b, const int* list, int nlist, const RedWsX& ws,
          double* out, const unsigned char* stc, const int* pos, double* out_c, double* out_c_dd)
{
    if (ncols < 1 || ncols > kColsX || (out_c != nullptr) != (out_c_dd != nullptr))
        return LBFGSX_E_INVALID;
    // byte model: a sector per column and vector at every listed row (out_c: cF too)
    model_add(double(nlist) * (64.0 * (ncols + 3 + (out_c ? 1 : 0)) + 4 + 64));
    // a short list: few blocks keep the reduction tail short
#define CALL(NCL, G)                                                                                                      \
    if (out_c)                                                                                                            \
        LBFGSX_LAUNCH((kx_list2<T, NCL, G, true>), dim3(std::min(32, grid_rows(nlist, 64 / G, 1, num_cus))), dim3(kBlock), 0, s, \
                      cols, ncols, b, list, nlist, ws, out, stc, pos, out_c, out_c_dd);                                   \
    else                                                                                                                  \
        LBFGSX_LAUNCH((kx_list2<T, NCL, G, false>), dim3(std::min(32, grid_rows(nlist, 64 / G, 1, num_cus))), dim3(kBlock), 0, s, \
                      cols, ncols, b, list, nlist, ws, out, stc, pos, out_c, out_c_dd)
    LBFGSX_XCLASS(ncols, CALL);
#undef CALL
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

template <class T>
int list1(hipStream_t s, int num_cus, const ColsX<T>& cols, int ncols, const BVecs<T>& b, int vsel_id, int mask, const int* list, int nlist,
          const RedWsX& ws, double* out, const unsigned char* stc, const int* pos, double* out_dd)
{
    if (ncols < 1 || ncols > kColsX)
        return LBFGSX_E_INVALID;
    model_add(double(nlist) * (64.0 * (ncols + 1) + 4 + 64));  // byte model: as list2, one vector
#define CALL(NCL, G)                                                                                                      \
    LBFGSX_LAUNCH((kx_list1<T, NCL, G>), dim3(std::min(32, grid_rows(nlist, 64 / G, 1, num_cus))), dim3(kBlock), 0, s, cols,   \
                  ncols, b, vsel_id, mask, list, nlist, ws, out, stc, pos, out_dd)
    LBFGSX_XCLASS(ncols, CALL);
#undef CALL
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

template <class T>
int multidot_mask(hipStream_t s, int num_cus, const ColsX<T>& cols, int ncols, const BVecs<T>& b, int vsel_id, const T* vcol, int mask,
                  int64_t n, const RedWsX& ws, double* out)
{
    if (ncols < 1 || ncols > kColsX)
        return LBFGSX_E_INVALID;
    const int64_t want = (n + int64_t(kWaves) * 256 - 1) / (int64_t(kWaves) * 256);
    model_add(double(n));  // byte model: the state bytes (the rows inside the mask are 10^1..10^4 of n)
#define CALL(NCL, G)                                                                                                   \
    LBFGSX_LAUNCH((kx_multidot_mask<T, NCL, G>),                                                                        \
                  dim3(int(std::max<int64_t>(1, std::min<int64_t>(want, std::min(occ_mask_x(NCL) * num_cus, kMaxGridX))))), \
                  dim3(kBlock), 0, s, cols, ncols, b, vsel_id, vcol, mask, n, ws, out)
    LBFGSX_XCLASS(ncols, CALL);
#undef CALL
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

template <class T>
int wf_append(hipStream_t s, const ColsX<T>& orig, int ncols, T* wf, int64_t wf_ld, int* wf_idx, int* pos, const int* enter,
              unsigned* cnt, unsigned cap, unsigned wf_cap, int split, int gap)
{
    LBFGSX_LAUNCH((kx_wf_append<T>), dim3(16), dim3(kBlock), 0, s, orig, ncols, wf, wf_ld, wf_idx, pos, enter, cnt, cap, wf_cap, split,
                  gap);
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

int gram_kpb(int ntot)
{
    const int npairs = ntot * (ntot + 1) / 2;
    const int kp = (npairs + 255) / 256;
    return kp <= 1 ? 1 : kp <= 2 ? 2 : kp <= 3 ? 3 : kp <= 4 ? 4 : kp <= 6 ? 6 : kp <= 9 ? 9 : 13;
}

template <class T, int KPB>
static int gram_kp(hipStream_t s, int max_blocks, const ColsX<T>& cols, int ncols, const BVecs<T>& b, int vsel_id, int mask,
                   int64_t n, double* partial, const ProX<T>& pro, const GramRows<T>& gr, double* fin_out, double* fin_dd,
                   unsigned long long* done, unsigned long long seq, unsigned* ticket)
{
    const int ntot = ncols + (vsel_id >= 0 ? 1 : 0);
    const int cs = ntot | 1;  // odd row stride: the lanes of a wave that read one row hit distinct banks
    const size_t lds = size_t(64) * size_t(cs) * sizeof(double) + 64 * sizeof(int);
    if (lds > 48 * 1024)  // per device and cheap: not cached
        (void) hipFuncSetAttribute(reinterpret_cast<const void*>(kx_gram<T, KPB>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    const int64_t nbatch = (n + 63) / 64;
    const int blocks = int(std::max<int64_t>(1, std::min<int64_t>(max_blocks, nbatch)));
    {
        // byte model: a list of rows -> a sector per column at every row; a masked pass over n rows -> the state bytes, the row
        // numbers and the columns of the rows kept (counted as all of them: an upper bound the full passes of the first
        // iterations reach), plus the compact copy it writes
        const double e = sizeof(T);
        if (gr.w_by_row && gr.in_idx)
            model_add(double(n) * (64.0 * ncols + 4 + 64));
        else
            model_add(double(n) * (double(ncols) * e * (gr.out_w ? 2 : 1) + 1 + (gr.in_idx ? 4 : 0) + (vsel_id >= 0 ? e : 0) +
                                   (gr.out_w ? 8 : 0)));
    }
    LBFGSX_LAUNCH((kx_gram<T, KPB>), dim3(blocks), dim3(kBlock), lds, s, cols, ncols, b, vsel_id, mask, n, partial, pro, gr, cs,
                  (blocks <= kGramSelfFinish && ticket) ? fin_out : static_cast<double*>(nullptr), fin_dd, done, seq, ticket);
    if (hipGetLastError() != hipSuccess)
        return -1;
    return blocks;
}

template <class T>
int gram(hipStream_t s, int max_blocks, const ColsX<T>& cols, int ncols, const BVecs<T>& b, int vsel_id, int mask, int64_t n,
         double* partial, const ProX<T>& pro, const GramRows<T>& gr, double* fin_out, double* fin_dd, unsigned long long* done,
         unsigned long long seq, unsigned* ticket)
{
    if (ncols < 1 || ncols > kColsX)
        return -1;
#define GK(K) return gram_kp<T, K>(s, max_blocks, cols, ncols, b, vsel_id, mask, n, partial, pro, gr, fin_out, fin_dd, done, seq, ticket)
    switch (gram_kpb(ncols + (vsel_id >= 0 ? 1 : 0)))
    {
    case 1: GK(1);
    case 2: GK(2);
    case 3: GK(3);
    case 4: GK(4);
    case 6: GK(6);
    case 9: GK(9);
    default: GK(13);
    }
#undef GK
}

int gram_finish(hipStream_t s, const double* partial, int blocks, int ntile, double* partial2, double* out, double* out_dd,
                unsigned long long* done, unsigned long long seq, unsigned* ticket)
{
    const int nch = std::min(blocks, 32);
    if (blocks > 1)
    {
        LBFGSX_LAUNCH(kx_gram_finish, dim3(ntile, nch), dim3(kBlock), 0, s, partial, blocks, partial2, 0, static_cast<double*>(nullptr),
                      static_cast<unsigned long long*>(nullptr), 0ull, ticket);
        LBFGSX_LAUNCH(kx_gram_finish, dim3(ntile, 1), dim3(kBlock), 0, s, partial2, nch, out, 1, out_dd, done, seq, ticket);
    }
    else
        LBFGSX_LAUNCH(kx_gram_finish, dim3(ntile, 1), dim3(kBlock), 0, s, partial, 1, out, 1, out_dd, done, seq, ticket);
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

#define INST(T)                                                                                                                  \
    template int rows<T>(hipStream_t, int, int, const ColsX<T>&, int, const BVecs<T>&, int, int, int64_t, const RedWsX&, double*,    \
                         double*, const ProX<T>&, const RowsX<T>&, int, int);                                                    \
    template int solve_sweep<T>(hipStream_t, int, int, const ColsX<T>&, int, const BVecs<T>&, const BVecs<T>&, int, const CoefX<T>&, \
                                int, T, int64_t, const RedWsX&, double*, int*, unsigned*, unsigned, const int*, T*, T*, int,     \
                                const ProX<T>*);                                                                                 \
    template int multidot2_wf<T>(hipStream_t, int, const ColsX<T>&, int, int, int, const T*, const T*, const T*, const int*, int64_t, \
                                 const ColsX<T>&, const int*, int, const RedWsX&, double*, T*, T*);                              \
    template int multidot2<T>(hipStream_t, int, const ColsX<T>&, int, const T*, const T*, int64_t, const RedWsX&, double*);        \
    template int list2<T>(hipStream_t, int, const ColsX<T>&, int, const BVecs<T>&, const int*, int, const RedWsX&, double*,         \
                          const unsigned char*, const int*, double*, double*);                                                   \
    template int list1<T>(hipStream_t, int, const ColsX<T>&, int, const BVecs<T>&, int, int, const int*, int, const RedWsX&,         \
                          double*, const unsigned char*, const int*, double*);                                                   \
    template int multidot_mask<T>(hipStream_t, int, const ColsX<T>&, int, const BVecs<T>&, int, const T*, int, int64_t,            \
                                  const RedWsX&, double*);                                                                       \
    template int gram<T>(hipStream_t, int, const ColsX<T>&, int, const BVecs<T>&, int, int, int64_t, double*, const ProX<T>&,      \
                         const GramRows<T>&, double*, double*, unsigned long long*, unsigned long long, unsigned*);                                                                                    \
    template int wf_append<T>(hipStream_t, const ColsX<T>&, int, T*, int64_t, int*, int*, const int*, unsigned*, unsigned, unsigned, int, \
                              int)
INST(double);
INST(float);
#undef INST

}  // namespace xl
}  // namespace lbfgsx
