// lbfgspp_amd/csrc/gram_i8.cuh -- K6 on the matrix cores, exactly: the masked Gram of solve_PtBP
// (/root/reference/include/LBFGSpp/BFGSMat.h:543-556,  W_P'W_P  column by column) as an error-free integer contraction.
//
// Why integers.  The parity contract wants every Gram entry to be the correctly rounded value of the exact sum (the
// oracle accumulates in extended precision), and ill-conditioned `mid` systems amplify even 1 ulp beyond 1e-10.  A
// floating-point MFMA rounds after every accumulate; v_mfma_i32_32x32x32_i8 does not round at all.  So each column is put
// on a fixed-point grid of its own (86 bits below a power of two that bounds the column, taken from the exact max |.| kept
// per history column) and cut into D = 11 signed radix-256 digits:
//     x  ~  2^(E - 86) * sum_k d_k 256^k ,   d_k in [-128, 127],  k = 0..10                (truncation < 2^(E - 86))
// A product of two elements is the sum of 121 digit products d_k d'_l 256^(k+l); the 66 with k + l >= 10 are kept (the
// rest is below 2^-82 of the two column scales), grouped by u = k + l - 10 into 11 int32 accumulator tiles: one
// v_mfma_i32_32x32x32_i8 per digit pair adds the contributions of 32 rows to all 32 x 32 column pairs at once.  Sums of
// integers are exact and order independent, so the result does not depend on the grid or on timing: per-wave int64
// partials, an integer tree, and ONE rounding at the very end (k_gram_i8_final: double-double assembly of
// sum_u V_u 256^(10+u), scaled by 2^(E_i + E_j - 172)).  Error against the exact sum: < 2^-80 of sum |x_i||x_j| for
// entries whose columns are used to within 2^-3 of their max -- far inside the half-ulp, like the double-double kernel
// (k_gram_dd) whose results it reproduces bit for bit in the tests.
//
// Layout.  Rows are loaded as in k_gram_dd: a wavefront takes 64 consecutive rows (lane = row, coalesced 512-byte column
// segments), drops the rows outside the mask by ballot compaction and stages the survivors in a wave-private LDS tile
// [row][col]; the prologue statement and the v row (W_P'v and v'v: 2c+1 double-double sums per lane, v is produced on
// the fly and has no known scale) are evaluated right there, lane = row.  For the matrix cores the tile is re-read in
// operand layout: lane l holds column l & 31 and the 16 rows of half l >> 5, cuts its 16 elements into digits and
// byte-transposes them into eleven 16-byte operands (one per digit); A and B operands of G = W'W are the same registers.
// MFMA: 66 instructions per 32 staged rows; VALU (digits): ~36 integer instructions per element.
#pragma once
#include "lbfgsb_kernels.cuh"

namespace lbfgsx {

constexpr int kI8Digits = 11;
constexpr int kI8Acc = 11;         // u = k + l - 10 = 0..10
constexpr int kI8Ring = 128;       // rows of the wave-private staging ring (31 left over + 64 new < 128)
constexpr int kI8FlushBatches = 120;  // 64-row batches between flushes: <= 241 groups, 241 * 32 * 11 * 2^14 < 2^31

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

struct GramI8Args
{
    const unsigned long long* colmax;  // bit patterns of max |column| per physical column: [0, m] Y, [m+1, 2m+1] S
    int cidx[32];                      // logical column -> index into colmax
    // The compact copy of the free rows (GramRows of lbfgsb_kernels.cuh), written on the way by a pass over the
    // full-length columns: row kept at position pos of batch bt goes to position out_base[bt] + pos.  Null: not written.
    double* out_w;
    int64_t out_ld;
    int out_split, out_gap;  // slot-stable columns: GramRows::out_split
    int* out_idx;
    const int* out_base;
    int* out_pos;
};

// 11 signed digits of t = trunc(x * 2^(86 - E)), E = emax - 1022 (emax: biased exponent of the column's max), as three
// words: bytes 0..3, 4..7, 8..10.  The 86-bit magnitude is peeled off in f64, 22 + 32 + 32 bits (every step exact: a
// truncation, the subtraction of an integer part, a multiplication by 2^32), which costs 11 full-rate instructions where
// shifting the 53-bit mantissa into a 96-bit field costs a dozen quarter-rate 64-bit shifts; the sign (two's complement)
// and the digit bias then take 10 integer instructions on the three words.
__device__ __forceinline__ void gram_i8_digits(double x, int lsh /* 1044 - emax */, unsigned& w0, unsigned& w1, unsigned& w2)
{
    const double a = __builtin_ldexp(__builtin_fabs(x), lsh);  // |t| / 2^64 < 2^22
    const double p2 = __builtin_trunc(a);
    const double bq = __builtin_ldexp(a - p2, 32);
    const double p1 = __builtin_trunc(bq);
    const double cq = __builtin_ldexp(bq - p1, 32);
    unsigned u2 = (unsigned) p2, u1 = (unsigned) p1, u0 = (unsigned) __builtin_trunc(cq);
    // two's complement for negative x, then + 0x80 in bytes 0..9 (all digits but the leading one): one carry chain
    const unsigned sm = unsigned(int(__double2hiint(x)) >> 31);  // all ones when negative
    u0 ^= sm;
    u1 ^= sm;
    u2 ^= sm;
    const unsigned long long s0 = (unsigned long long) u0 + (0x80808080u - sm);  // - sm = + 1 when negative
    const unsigned long long s1 = (unsigned long long) u1 + 0x80808080u + (s0 >> 32);
    u2 = u2 + 0x00008080u + unsigned(s1 >> 32);
    w0 = unsigned(s0) ^ 0x80808080u;
    w1 = unsigned(s1) ^ 0x80808080u;
    w2 = u2 ^ 0x00008080u;
}

// 4 x 4 byte transpose: word w of four elements -> four words, byte j of output k = byte k of input j
__device__ __forceinline__ void gram_i8_tr4(unsigned a, unsigned b, unsigned c, unsigned d, unsigned (&o)[4])
{
    const unsigned t0 = __builtin_amdgcn_perm(b, a, 0x05010400u);  // a0 b0 a1 b1
    const unsigned t1 = __builtin_amdgcn_perm(b, a, 0x07030602u);  // a2 b2 a3 b3
    const unsigned t2 = __builtin_amdgcn_perm(d, c, 0x05010400u);  // c0 d0 c1 d1
    const unsigned t3 = __builtin_amdgcn_perm(d, c, 0x07030602u);  // c2 d2 c3 d3
    o[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);             // a0 b0 c0 d0
    o[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);             // a1 b1 c1 d1
    o[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);             // a2 b2 c2 d2
    o[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);             // a3 b3 c3 d3
}

// entry (i, j), i >= j, of the packed lower triangle
__device__ __forceinline__ int gram_tri(int i, int j) { return i * (i + 1) / 2 + j; }

// part_i: [blocks][kI8Acc][ne_pad] int64, zeroed by the host (ne_pad = padded number of lower-triangle entries of the
// ncols x ncols block)
// part_v: [waves][32][2] double-double sums of the v row: entry j < ncols = v . col_j, entry ncols = v . v
template <int CS>
__global__ void __launch_bounds__(kBlock, 1)
    k_gram_i8(Cols<double, 32> cols, int ncols, BVecs<double> b, int vsel_id, int mask, int64_t n,
              long long* __restrict__ part_i, int ne_pad, double* __restrict__ part_v, GramPrologue<double> pro, GramI8Args ga,
              const int* __restrict__ ridx)
{
    // ridx: `cols` is the compact copy of the free rows (GramRows): n of them, row t of the columns = row ridx[t] of the
    // vectors and of the state bytes; null: the full-length columns, row t = row t
    constexpr int cs = CS;
    extern __shared__ double tile[];
    __shared__ double pc1[64], pc2[64];
    __shared__ int s_emax[32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 64)
    {
        pc1[tid] = pro.c1[tid];
        pc2[tid] = pro.c2[tid];
    }
    if (tid < 32)
        s_emax[tid] = (tid < ncols) ? int((ga.colmax[ga.cidx[tid]] >> 52) & 0x7FFull) : 0;
    __syncthreads();
    double* tl = tile + wv * (kI8Ring * cs);
    const int mc = lane & 31, mh = lane >> 5;  // operand layout: column, row half
    const int my_lsh = 1044 - s_emax[mc];
    const bool col_ok = mc < ncols;
    const int64_t gwave = int64_t(blockIdx.x) * (kBlock / 64) + wv;
    const int64_t nwaves = int64_t(gridDim.x) * (kBlock / 64);
    long long* mypart = part_i + int64_t(blockIdx.x) * int64_t(kI8Acc) * ne_pad;

    i32x16 acc[kI8Acc];
#pragma unroll
    for (int u = 0; u < kI8Acc; u++)
#pragma unroll
        for (int r = 0; r < 16; r++)
            acc[u][r] = 0;
    // v row: lane l accumulates entry l % (ncols + 1) -- v . col_j for j < ncols, v . v for j = ncols -- over the staged
    // rows l / (ncols + 1), + nvg, ... (nvg = 64 / (ncols + 1) rows per trip use all lanes); v sits in tile column ncols
    DD accv;
    const int nv1 = ncols + 1, nvg = 64 / nv1;
    const int vj = lane % nv1, vg = lane / nv1;

    const int64_t nbatch = (n + kGramDDRows - 1) / kGramDDRows;
    // the row of the vectors behind position (bq, lane) of the columns
    auto row_of = [&](int64_t bq) -> int64_t {
        const int64_t rq = bq * kGramDDRows + lane;
        if (!(bq < nbatch && rq < n))
            return 0;
        return ridx ? int64_t(ridx[rq]) : rq;
    };
    auto load_st = [&](int64_t bq, int64_t row) -> unsigned char {
        const int64_t rq = bq * kGramDDRows + lane;
        return (mask && bq < nbatch && rq < n) ? b.st[row] : (unsigned char) 0;
    };
    auto keep_of = [&](int64_t bq, unsigned char stq) -> bool {
        const int64_t rq = bq * kGramDDRows + lane;
        return bq < nbatch && rq < n && (!mask || (stq & mask));
    };
    const bool need_rhs = pro.mode == GP_RHS || vsel_id == VS_NEG_RHS;
    const bool need_g = pro.mode == GP_LINEAR;

    // ---- software pipeline.  State bytes run two batches ahead; the column values (vn) and the per-row inputs of the
    // prologue / of v (an*) one batch ahead: they are requested right after the previous batch has been staged, so the
    // contraction of that batch -- the long part -- covers their latency (one wavefront per SIMD: nothing else would).
    // (with a row list the row numbers run three batches ahead, one ahead of the state bytes they address)
    int64_t bt = gwave;
    int64_t row_a = row_of(bt), row_b = row_of(bt + nwaves), row_c = row_of(bt + 2 * nwaves);
    unsigned char st_a = load_st(bt, row_a), st_b = load_st(bt + nwaves, row_b);
    bool keep_n = keep_of(bt, st_a);
    int64_t row_n = row_a;  // row of the batch whose values are in flight (vn, an*)
    double vn[CS], an0 = 0.0, an1 = 0.0, an2 = 0.0;
    auto issue_loads = [&](int64_t bq, bool kq, int64_t row) {
        const int64_t cq = kq ? bq * kGramDDRows + lane : int64_t(0);  // masked-out lanes re-read row 0 (no branch)
        const int64_t rq = kq ? row : int64_t(0);
#pragma unroll
        for (int j = 0; j < CS; j++)
            vn[j] = cols.p[j < ncols ? j : ncols - 1][cq];              // columns beyond ncols repeat the last one
        an0 = need_rhs ? b.rhs[rq] : (need_g ? b.g[rq] : 0.0);
        an1 = an2 = 0.0;
        switch (vsel_id)
        {
        case VS_DRT: an1 = b.drt[rq]; break;
        case VS_NEG_CF: if (!need_g) an1 = b.cF[rq]; break;
        case VS_LBOUND: an1 = b.lb[rq]; an2 = b.x0[rq]; break;
        case VS_UBOUND: an1 = b.ub[rq]; an2 = b.x0[rq]; break;
        case VS_Y: an1 = b.y[rq]; break;
        default: break;
        }
    };
    issue_loads(bt, keep_n, row_n);

    int head = 0, fill = 0;  // ring rows [head, head + fill) are staged and not yet contracted
    bool last = false;
    for (;;)
    {
        // ---- flush interval: at most kI8FlushBatches batches (2 groups each at most) between two flushes
        for (int it = 0; it < kI8FlushBatches && !last; it++)
        {
            if (bt < nbatch)
            {
                // -- stage the batch whose values have arrived
                const int64_t r = row_n;  // row of the vectors (= bt * kGramDDRows + lane without a row list)
                const bool keep = keep_n;
                const unsigned long long bal = __ballot(keep);
                const int cnt = __popcll(bal);
                const int pos = __popcll(bal & ((1ull << lane) - 1ull));
                const int base = head + fill;
                if (keep)
                {
                    double* row = tl + ((base + pos) & (kI8Ring - 1)) * cs;
#pragma unroll
                    for (int j = 0; j < CS; j++)
                        row[j] = vn[j];
                    if (ga.out_w)  // the dense copy of the free rows for the passes that follow (ridx is null here)
                    {
                        const int64_t ot = int64_t(ga.out_base[bt]) + pos;
                        ga.out_idx[ot] = int(r);
                        ga.out_pos[r] = int(ot);
#pragma unroll
                        for (int j = 0; j < CS; j++)
                            if (j < ncols)
                                ga.out_w[int64_t(j + (j >= ga.out_split ? ga.out_gap : 0)) * ga.out_ld + ot] = vn[j];
                    }
                    double rhs_new = an0, cF_new = an1;
                    if (pro.mode != GP_NONE)
                    {
                        // (W * coef)(row): columns in order, plain accumulation -- the statement k_wcombine evaluates
                        double a1 = 0.0, a2 = 0.0;
                        if (pro.use1)
                        {
#pragma unroll
                            for (int j = 0; j < CS; j++)
                                if (j < ncols)
                                    a1 = a1 + vn[j] * pc1[j];
                        }
                        if (pro.use2)
                        {
#pragma unroll
                            for (int j = 0; j < CS; j++)
                                if (j < ncols)
                                    a2 = a2 + vn[j] * pc2[j];
                        }
                        if (pro.mode == GP_RHS)
                        {
                            double rh = an0;
                            if (pro.use1)
                                rh = rh + (-a1);
                            if (pro.use2)
                                rh = rh + (-a2);
                            b.rhs[r] = rh;
                            rhs_new = rh;
                        }
                        else
                        {
                            cF_new = (pro.use1 ? (-1.0 * a1) : 0.0) + an0;
                            b.cF[r] = cF_new;
                        }
                    }
                    if (vsel_id >= 0)
                    {
                        double vr;  // vsel() of lbfgsb_kernels.cuh on the values already in registers
                        switch (vsel_id)
                        {
                        case VS_NEG_CF: vr = -cF_new; break;
                        case VS_NEG_RHS: vr = -rhs_new; break;
                        case VS_LBOUND: vr = an1 - an2; break;
                        case VS_UBOUND: vr = an1 - an2; break;
                        default: vr = an1; break;  // VS_DRT, VS_Y
                        }
                        row[ncols] = vr;  // after the columns: column ncols of the tile is v
                    }
                }
                // -- request the next batch
                const int64_t bn = bt + nwaves;
                st_a = st_b;
                st_b = load_st(bn + nwaves, row_c);
                row_n = row_b;
                row_b = row_c;
                row_c = row_of(bn + 2 * nwaves);
                keep_n = keep_of(bn, st_a);
                issue_loads(bn, keep_n, row_n);
                bt = bn;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (vsel_id >= 0 && vg < nvg)
                    for (int rr = vg; rr < cnt; rr += nvg)
                    {
                        const double* rw = tl + ((base + rr) & (kI8Ring - 1)) * cs;
                        accv.add_prod(rw[ncols], rw[vj]);
                    }
                fill += cnt;
            }
            else
                last = true;
            // ---- groups of 32 staged rows (the last one zero-padded) onto the matrix cores: the digits of the group at
            // the head of the ring are cut (VALU), then its 66 MFMAs are issued.  ONE copy of this code: with a second
            // inlined copy the accumulator tiles went to scratch memory.
            while (fill >= 32 || (last && fill > 0))
            {
                const int nrows = fill < 32 ? fill : 32;
                i32x4 dcur[kI8Digits];
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    unsigned w0[4], w1[4], w2[4];
#pragma unroll
                    for (int t = 0; t < 4; t++)
                    {
                        const int rr = 16 * mh + 4 * q + t;
                        const double x = (col_ok && rr < nrows) ? tl[((head + rr) & (kI8Ring - 1)) * cs + mc] : 0.0;
                        gram_i8_digits(x, my_lsh, w0[t], w1[t], w2[t]);
                    }
                    unsigned o[4];
                    gram_i8_tr4(w0[0], w0[1], w0[2], w0[3], o);
                    dcur[0][q] = int(o[0]);
                    dcur[1][q] = int(o[1]);
                    dcur[2][q] = int(o[2]);
                    dcur[3][q] = int(o[3]);
                    gram_i8_tr4(w1[0], w1[1], w1[2], w1[3], o);
                    dcur[4][q] = int(o[0]);
                    dcur[5][q] = int(o[1]);
                    dcur[6][q] = int(o[2]);
                    dcur[7][q] = int(o[3]);
                    gram_i8_tr4(w2[0], w2[1], w2[2], w2[3], o);
                    dcur[8][q] = int(o[0]);
                    dcur[9][q] = int(o[1]);
                    dcur[10][q] = int(o[2]);
                }
                // 66 digit pairs, accumulator u collects k + l = 10 + u; round-robin over the
                // accumulators so that consecutive MFMAs are independent
#pragma unroll
                for (int k = 0; k < kI8Digits; k++)
#pragma unroll
                    for (int u = 0; u < kI8Acc; u++)
                    {
                        const int l = 10 + u - k;
                        if (l >= 0 && l < kI8Digits)
                            acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(dcur[k], dcur[l], acc[u], 0, 0, 0);
                    }
                head = (head + nrows) & (kI8Ring - 1);
                fill -= nrows;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- flush the int32 tiles into the block's int64 partials (zeroed by the host): integer atomics -- exact, order
        // independent, and no read-modify-write through registers (176 of those at once were what spilled).  C/D layout
        // of the 32x32 tile: column j = lane & 31, row i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int u = 0; u < kI8Acc; u++)
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * mh, j = mc;
                if (i < ncols && j <= i && acc[u][r] != 0)
                    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(mypart + int64_t(u) * ne_pad + gram_tri(i, j)),
                                           (unsigned long long) (long long) acc[u][r], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                acc[u][r] = 0;
            }
        if (last)
            break;
    }
    // v row: the lanes that hold the same entry are nv1 apart
    if (vsel_id >= 0)
    {
        __builtin_amdgcn_wave_barrier();
        double* scr = tl;  // the tile is free now: [64][2]
        scr[lane * 2 + 0] = accv.hi;
        scr[lane * 2 + 1] = accv.lo;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < nv1)
        {
            DD t;
            for (int g = 0; g < nvg; g++)
                t.merge(scr[(g * nv1 + lane) * 2 + 0], scr[(g * nv1 + lane) * 2 + 1]);
            part_v[(gwave * 32 + lane) * 2 + 0] = t.hi;
            part_v[(gwave * 32 + lane) * 2 + 1] = t.lo;
        }
    }
}

// integer tree, level 1: sum the per-wave partials.  grid = (kI8Acc, chunks); vsum[u][e] (int64) via atomics on the few
// chunk results (integer addition: exact, order independent)
__global__ void __launch_bounds__(kBlock) k_gram_i8_sum(const long long* __restrict__ part_i, int nwaves, int ne, int ne_pad,
                                                        unsigned long long* __restrict__ vsum)
{
    const int u = blockIdx.x, ch = blockIdx.y, nch = gridDim.y;
    for (int e = threadIdx.x; e < ne; e += kBlock)
    {
        long long t = 0;
        for (int w = ch; w < nwaves; w += nch)
            t += part_i[(int64_t(w) * kI8Acc + u) * ne_pad + e];
        atomicAdd(vsum + u * ne_pad + e, (unsigned long long) t);
    }
}

// the one rounding: entry e = (i, j) of the ncols x ncols block from its 11 integer sums; the v row from the per-wave
// double-double partials.  out / out_dd in the packed lower-triangle format of k_gram_finish's callers:
// e = i (i + 1) / 2 + j over ntot = ncols (+ 1 with v) rows.
__global__ void __launch_bounds__(kBlock) k_gram_i8_final(const unsigned long long* __restrict__ vsum, int ncols, int ne_pad,
                                                          const double* __restrict__ part_v, int nwaves, int with_v,
                                                          GramI8Args ga, double* __restrict__ out, double* __restrict__ out_dd)
{
    const int ne = ncols * (ncols + 1) / 2;
    for (int e = threadIdx.x; e < ne; e += kBlock)
    {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= e)
            i++;
        const int j = e - i * (i + 1) / 2;
        const int ei = int((ga.colmax[ga.cidx[i]] >> 52) & 0x7FFull) - 1022;
        const int ej = int((ga.colmax[ga.cidx[j]] >> 52) & 0x7FFull) - 1022;
        DD t;
        for (int u = kI8Acc - 1; u >= 0; u--)
        {
            const long long v = (long long) vsum[u * ne_pad + e];
            const double hi = double(v);                       // |v| < 2^56: the difference below is exact
            const double lo = double(v - (long long) hi);
            const int sc = ei + ej - 172 + 8 * (10 + u);
            t.add(ldexp(hi, sc));
            t.add(ldexp(lo, sc));
        }
        const double s = t.hi + t.lo;  // renormalise
        t.lo = t.lo - (s - t.hi);
        t.hi = s;
        out[e] = t.value();
        if (out_dd)
        {
            out_dd[e * 2 + 0] = t.hi;
            out_dd[e * 2 + 1] = t.lo;
        }
    }
    if (with_v)
    {
        // v row: 32 entries x 8 slices of the per-wave partials per thread, then the 8 slices of an entry in order
        __shared__ double sv[8][32][2];
        const int j = threadIdx.x & 31, sl = threadIdx.x >> 5;
        DD t;
        for (int w = sl; w < nwaves; w += 8)
            t.merge(part_v[(int64_t(w) * 32 + j) * 2 + 0], part_v[(int64_t(w) * 32 + j) * 2 + 1]);
        sv[sl][j][0] = t.hi;
        sv[sl][j][1] = t.lo;
        __syncthreads();
        if (threadIdx.x <= ncols)
        {
            DD r;
            for (int q = 0; q < 8; q++)
                r.merge(sv[q][threadIdx.x][0], sv[q][threadIdx.x][1]);
            out[ne + threadIdx.x] = r.value();
            if (out_dd)
            {
                out_dd[(ne + threadIdx.x) * 2 + 0] = r.hi;
                out_dd[(ne + threadIdx.x) * 2 + 1] = r.lo;
            }
        }
    }
}

// exact max |col| of one column into its slot (columns that did not come through k_b_post)
template <class T>
__global__ void __launch_bounds__(kBlock) k_colmax2(const T* __restrict__ s, const T* __restrict__ y, int64_t n,
                                                    unsigned long long* slot_s, unsigned long long* slot_y)
{
    double ms = 0.0, my = 0.0;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        ms = fmax(ms, fabs(double(s[i])));
        my = fmax(my, fabs(double(y[i])));
    }
    block_atomic_max(slot_s, ms);
    block_atomic_max(slot_y, my);
}

}  // namespace lbfgsx
