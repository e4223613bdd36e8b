// lbfgspp_amd/csrc/lbfgsb_x.hpp -- launchers of the kernels of lbfgsb_x.cuh (defined and instantiated in lbfgsb_x.hip, a
// translation unit of its own: 8 column classes x 2 element types x 7 kernels compile next to lbfgsb.hip, not inside it).
// Every launcher picks the class (NCL columns per lane, G lanes per row) from 2c and the grid from the rows.
#pragma once
#include "ctx.hpp"
#include "lbfgsb_x.cuh"

namespace lbfgsx {
namespace xl {

template <class T>
int rows(hipStream_t s, int num_cus, int na, const ColsX<T>& cols, int ncols, const BVecs<T>& b, int vsel_id, int mask, int64_t n,
         const RedWsX& ws, double* out, double* out_dd, const ProX<T>& pro, const RowsX<T>& gr, int col_a, int col_b);
template <class T>
int solve_sweep(hipStream_t s, int num_cus, int first, const ColsX<T>& cols, int ncols, const BVecs<T>& b, const BVecs<T>& bw,
                int vsel_id, const CoefX<T>& coef, int has_w, T theta, int64_t n, const RedWsX& ws, double* out, int* lu_list,
                unsigned* lu_cnt, unsigned lu_cap, const int* ridx, T* cli, T* cui, int cv, const ProX<T>* pro = nullptr);
template <class T>
int multidot2_wf(hipStream_t s, int num_cus, const ColsX<T>& wfc, int ncols, int fresh_a, int fresh_b, const T* snew, const T* ynew,
                 const T* dvec, const int* idx, int64_t npos, const ColsX<T>& full, const int* list, int nlist, const RedWsX& ws,
                 double* out, T* dst_a = nullptr, T* dst_b = nullptr);
template <class T>
int multidot2(hipStream_t s, int num_cus, const ColsX<T>& cols, int ncols, const T* v1, const T* v2, int64_t n, const RedWsX& ws,
              double* out);
template <class T>
int list2(hipStream_t s, int num_cus, const ColsX<T>& cols, int ncols, const BVecs<T>& b, const int* list, int nlist, const RedWsX& ws,
          double* out, const unsigned char* stc, const int* pos, double* out_c = nullptr, double* out_c_dd = nullptr);
template <class T>
int list1(hipStream_t s, int num_cus, const ColsX<T>& cols, int ncols, const BVecs<T>& b, int vsel_id, int mask, const int* list, int nlist,
          const RedWsX& ws, double* out, const unsigned char* stc = nullptr, const int* pos = nullptr, double* out_dd = nullptr);
template <class T>
int multidot_mask(hipStream_t s, int num_cus, const ColsX<T>& cols, int ncols, const BVecs<T>& b, int vsel_id, const T* vcol, int mask,
                  int64_t n, const RedWsX& ws, double* out);
template <class T>
int wf_append(hipStream_t s, const ColsX<T>& orig, int ncols, T* wf, int64_t wf_ld, int* wf_idx, int* pos, const int* enter,
              unsigned* cnt, unsigned cap, unsigned wf_cap, int split, int gap);
// entries per thread of kx_gram for 2c + 1 (+ v) = ntot columns; the partial buffer holds [blocks][gram_kpb * 256][2] doubles
int gram_kpb(int ntot);
// returns the number of blocks launched, < 0: error.  More than kGramSelfFinish blocks (or no ticket word): their partials
// wait for gram_finish.  Up to kGramSelfFinish blocks (max_blocks: the Grams over the short row lists of the sweeps) with
// fin_out and a zeroed ticket word given: the launch itself leaves the rounded entries in fin_out, the (hi, lo) pairs in fin_dd
// and stores the completion word -- no gram_finish
template <class T>
int gram(hipStream_t s, int max_blocks, const ColsX<T>& cols, int ncols, const BVecs<T>& b, int vsel_id, int mask, int64_t n,
         double* partial, const ProX<T>& pro, const GramRows<T>& gr, double* fin_out = nullptr, double* fin_dd = nullptr,
         unsigned long long* done = nullptr, unsigned long long seq = 0, unsigned* ticket = nullptr);
// two-level sum of `blocks` partial sets of `ntile` tiles each: rounded entries to out[ntile * 256], (hi, lo) to out_dd
int gram_finish(hipStream_t s, const double* partial, int blocks, int ntile, double* partial2, double* out, double* out_dd,
                unsigned long long* done, unsigned long long seq, unsigned* ticket);

}  // namespace xl
}  // namespace lbfgsx
