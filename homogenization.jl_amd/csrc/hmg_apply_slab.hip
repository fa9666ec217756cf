// Operator apply for cells larger than the LDS (3D level 7: 47 905 nodes, 374 KiB), round 5: ONE persistent 1024-thread workgroup
// per CU whose waves have ROLES.  ref: src/apply_local_operators.jl:85-133 (+ :7-27 residual, constraint, the fused CG pass).
//
// k_apply_slab (hmg_kernels.hip) walks a cell in slabs of k-planes through a rolling LDS window and does, per slab, one thing after
// the other: rearrange the window, HBM -> LDS, evaluate -- two workgroups per CU overlap only by being out of step, at 64 VGPRs per
// thread, with every LDS read of the evaluation a round trip of its own (profiles/r05_level7_phase_timing.txt: rearranging 14-23 %,
// the HBM phase 28-60 %, the evaluation 25-51 % of a workgroup's life, one after the other).  Here:
//   * waves 0-7 are LOADERS: they fill window (g+1) & 1 -- the two halo planes from the other window (LDS -> LDS, no barrier: the
//     ranges are disjoint), the new planes from HBM as ONE stream of rows of 512 slots that runs across slabs and cells, the data
//     of a row requested S2_D rows and its addressing word 2 S2_D rows ahead of its use, with every load-phase side effect of the
//     fused CG pass (p-update, pending x-updates, r.r) exactly as k_apply_slab does them -- and the class weight table of the next cell;
//   * waves 8-15 are EVALUATORS: they evaluate step g from window g & 1 -- per node all 15 (surface: 2 x 8 + the weight row) LDS
//     reads issued back to back in ONE asm statement (128 VGPRs per thread are there for it: one workgroup per CU), source values
//     and addressing words requested a chunk (interior) or a whole step (surface) ahead;
//   * ONE workgroup barrier per slab, none inside it; a workgroup walks cells b, b + G, b + 2 G, .. (G = grid = CUs) and the
//     pipeline runs across cells: the loaders fill the first window of the next cell while the last slab of this one is evaluated.
// What the compiler must not be left to do here (measured, profiles/r05_experiments.txt): a store that sits between a load and its
// first use on SOME control-flow path makes the wait for that load a vmcnt(0) -- the store's round trip to L2, 2000 cycles, per
// node.  So every global store is an asm statement (not counted: the compiler's waits see loads only and stay exact or err towards
// one more load), and every global load is unconditional (rows are completed with copies of their last entry, whose results are
// thrown away).
// Same arithmetic per node in the same order as k_apply_slab: out, xout, xacc are equal to its results to the last bit
// (tests/test_gpu_slab2.py); the per-cell partial sums of p.Ap and r.r are summed in another, fixed, order.
#include "hmg_stencil.hpp"

namespace hmg {

namespace {

#ifdef HMG_PHASE_TIMING   // dev build: thread 0 (a loader) and thread 512 (an evaluator) add up their work and barrier-wait times
#define S2_T0() long long _tl = wall_clock64(); double _ph[6] = {0, 0, 0, 0, 0, 0}; const long long _tstart = _tl; long long _tb = _tl; double _ws = 0, _ws2 = 0, _wn = 0, _wmx = 0
// every wave: its time from a step barrier's exit to its arrival at the next one -- sum, sum of squares, count, maximum (tools/dev/slab2_wave_timing.py)
#define S2_ARRIVE() do { const double _d = (double)(wall_clock64() - _tb); _ws += _d; _ws2 += _d * _d; _wn += 1.0; _wmx = _d > _wmx ? _d : _wmx; } while (0)
#define S2_LEAVE() do { _tb = wall_clock64(); } while (0)
#define S2_STORE_WAVE() do { if (a.blockpart && (threadIdx.x & 63) == 0) { double *_o = a.blockpart + 2 * (size_t)a.nwork + 20 * (size_t)gridDim.x + 64 * (size_t)blockIdx.x + 4 * (threadIdx.x >> 6); _o[0] = _ws; _o[1] = _ws2; _o[2] = _wn; _o[3] = _wmx; } } while (0)
#define S2_PHASE(i)                          \
    do {                                     \
        const long long _n = wall_clock64(); \
        _ph[i] += (double)(_n - _tl);        \
        _tl = _n;                            \
    } while (0)
#define S2_STORE(base)                                                                   \
    do {                                                                                 \
        if (a.blockpart && rt == 0) {                                                    \
            double *_o = a.blockpart + 2 * (size_t)a.nwork + 16 * (size_t)blockIdx.x + (base); \
            _o[0] = (double)_tstart;                                                     \
            for (int _q = 0; _q < 4; ++_q) _o[1 + _q] = _ph[_q];                         \
            _o[5] = (double)wall_clock64();                                              \
            _o[6] = _ph[4];                                                              \
            _o[7] = _ph[5];                                                              \
        }                                                                                \
    } while (0)
// shader-clock cycles of the pieces of one interior node evaluation (every one of the first evaluator wave), summed: decode, LDS batch, arithmetic
#define S2_CYC_DECL() double _cy[4] = {0, 0, 0, 0}; long long _c0 = 0, _c1 = 0, _c2 = 0
#define S2_CYC(i) _c##i = (long long)__builtin_readcyclecounter()
#define S2_CYC_ADD(dep) do { asm volatile("" : "+v"(dep)); long long _c3 = (long long)__builtin_readcyclecounter(); _cy[0] += (double)(_c1 - _c0); _cy[1] += (double)(_c2 - _c1); _cy[2] += (double)(_c3 - _c2); _cy[3] += 1.0; } while (0)
#define S2_CYC_STORE() do { if (a.blockpart && rt == 0) { double *_o = a.blockpart + 2 * (size_t)a.nwork + 16 * (size_t)gridDim.x + 4 * (size_t)blockIdx.x; for (int _q = 0; _q < 4; ++_q) _o[_q] = _cy[_q]; } } while (0)
#else
#define S2_T0()
#define S2_ARRIVE()
#define S2_LEAVE()
#define S2_STORE_WAVE()
#define S2_PHASE(i)
#define S2_STORE(base)
#define S2_CYC_DECL()
#define S2_CYC(i)
#define S2_CYC_ADD(dep)
#define S2_CYC_STORE()
#endif

constexpr int S2_NT = 1024;        // threads per workgroup
// loader threads NL = 64 x NLW (the first NLW waves), evaluator threads NE = 1024 - NL: NLW = 4 (operator applies) or 8 (the restriction through the window: an eighth of the nodes is evaluated)
constexpr int S2_WROW = 16;        // doubles per class row of the weight table in LDS (128-B rows)
constexpr int S2_WSZ = 15 * S2_WROW;
constexpr int S2_RED = 16;         // per cell parity: 12 evaluator waves' p.Ap, 4 loader waves' r.r
constexpr int S2_TAB = 128;        // doubles (= 256 ints) of per-slab constants: 16 slabs x 16 ints
constexpr int S2_MAXSLAB = 16;
constexpr int S2_GUARD = 72;       // doubles of zeros below window 0: what the in-plane taps of plane 0's first rows address (up to m + 2 entries below the cell)
constexpr int S2_MAXCELLS = 1024;  // cells per workgroup whose ids are staged in LDS (more: the launch takes k_apply_slab)
__host__ __device__ constexpr int s2_smax(int nlw) { return nlw == 4 ? 4 : 6; }   // rows (of NE) of surface entries per slab (more: k_apply_slab)
__host__ __device__ constexpr int s2_umax(int nlw) { return nlw == 4 ? 8 : 4; }   // rows (of NE) of interior entries per slab (more: k_apply_slab)
// loader: a row is R x 256 slots -- the scalar bookkeeping of a row is paid once per R slots of a lane; D rows lie between the request
// of a row's data and its use (its addressing words: twice that).  NS streams of 8 B per slot, 256 lanes: 48 / 64 / 72 KB in flight
// per CU -- 25 / 49 / 49 KB, what 13 GB/s per CU (the share of 3.3 TB/s read) need at 2-4 us of loaded latency -- in 48-72 VGPRs of ring
__host__ __device__ constexpr int s2_rows(int ns, int nlw) { return nlw == 4 ? (ns == 1 ? 6 : 4) : 4; }
__host__ __device__ constexpr int s2_depth(int ns, int nlw) { return nlw == 4 ? 2 : ns == 1 ? 3 : 2; }

__device__ __forceinline__ uint32_t lds_addr(const double *p)
{
    return (uint32_t)(uintptr_t)(const lds_f64 *)p;
}

// 15 LDS reads of an interior node's stencil, issued back to back, ONE wait (form (i) of the guide's asm rules: loads and their
// s_waitcnt in one statement, early-clobber outputs).  ap = LDS byte address of the node.  Tap order of stencil_eval_v.
__device__ __forceinline__ void read15(uint32_t ap, int len, int A, int B, double (&v)[15])
{
    const uint32_t a0 = ap - 8u;                                   // p - 1 | p | p + 1
    const uint32_t a1 = ap + 8u * (uint32_t)(len - 1);             // p + len - 1 | p + len
    const uint32_t a2 = ap - 8u * (uint32_t)(len + 1);             // p - len - 1 | p - len
    const uint32_t pu = ap + 8u * (uint32_t)A, pd = ap - 8u * (uint32_t)B;
    const uint32_t a3 = pu - 8u * (uint32_t)len;                   // pu - len | pu + 1 - len
    const uint32_t a4 = pu - 8u;                                   // pu - 1 | pu
    const uint32_t a6 = pd + 8u * (uint32_t)len;                   // pd + len | pd + len + 1
    asm volatile("ds_read_b64 %0, %15 offset:8\n\t"
                 "ds_read_b64 %1, %15 offset:16\n\t"
                 "ds_read_b64 %2, %15\n\t"
                 "ds_read_b64 %3, %16\n\t"
                 "ds_read_b64 %4, %17 offset:8\n\t"
                 "ds_read_b64 %5, %16 offset:8\n\t"
                 "ds_read_b64 %6, %17\n\t"
                 "ds_read_b64 %7, %18\n\t"
                 "ds_read_b64 %8, %21 offset:8\n\t"
                 "ds_read_b64 %9, %19\n\t"
                 "ds_read_b64 %10, %20 offset:8\n\t"
                 "ds_read_b64 %11, %19 offset:8\n\t"
                 "ds_read_b64 %12, %20\n\t"
                 "ds_read_b64 %13, %18 offset:8\n\t"
                 "ds_read_b64 %14, %21\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
                   "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14])
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(pd), "v"(a6));
}

// Surface node: the interior node's addressing (seven address registers, read15's tap order) and, per statement, eight / seven
// taps' values and their class weights (the weight row: 128-B aligned, stride S2_WROW).  Taps that leave the cell carry zero
// weights (host check); what they read is some other entry of the window or the zero guards below / behind it -- finite values.
// A node of plane k = 0 has no plane below: the caller passes B = 0 (its four lower taps then read the node's own plane).
__device__ __forceinline__ double surface_eval(uint32_t ap, uint32_t aw, int len, int A, int B, double &ctr)
{
    const uint32_t a0 = ap - 8u;                                   // p - 1 | p | p + 1
    const uint32_t a1 = ap + 8u * (uint32_t)(len - 1);             // p + len - 1 | p + len
    const uint32_t a2 = ap - 8u * (uint32_t)(len + 1);             // p - len - 1 | p - len
    const uint32_t pu = ap + 8u * (uint32_t)A, pd = ap - 8u * (uint32_t)B;
    const uint32_t a3 = pu - 8u * (uint32_t)len;                   // pu - len | pu + 1 - len
    const uint32_t a4 = pu - 8u;                                   // pu - 1 | pu
    const uint32_t a6 = pd + 8u * (uint32_t)len;                   // pd + len | pd + len + 1
    double v[8], w[8];
    asm volatile("ds_read_b64 %0, %16 offset:8\n\t"
                 "ds_read_b64 %1, %16 offset:16\n\t"
                 "ds_read_b64 %2, %16\n\t"
                 "ds_read_b64 %3, %17\n\t"
                 "ds_read_b64 %4, %18 offset:8\n\t"
                 "ds_read_b64 %5, %17 offset:8\n\t"
                 "ds_read_b64 %6, %18\n\t"
                 "ds_read_b64 %7, %19\n\t"
                 "ds_read_b64 %8, %20\n\t"
                 "ds_read_b64 %9, %20 offset:8\n\t"
                 "ds_read_b64 %10, %20 offset:16\n\t"
                 "ds_read_b64 %11, %20 offset:24\n\t"
                 "ds_read_b64 %12, %20 offset:32\n\t"
                 "ds_read_b64 %13, %20 offset:40\n\t"
                 "ds_read_b64 %14, %20 offset:48\n\t"
                 "ds_read_b64 %15, %20 offset:56\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(w[0]),
                   "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(aw));
    // (the arithmetic of stencil_eval_c as hipcc compiles it: the product of tap 1 is rounded, tap 0 joins it by a fused multiply-add,
    //  then one fused multiply-add per tap in tap order)
    ctr = v[0];
    double acc = w[1] * v[1];
    acc = __builtin_fma(w[0], ctr, acc);
#pragma unroll
    for (int d = 2; d < 8; ++d) acc = __builtin_fma(w[d], v[d], acc);
    asm volatile("ds_read_b64 %0, %17 offset:8\n\t"
                 "ds_read_b64 %1, %15\n\t"
                 "ds_read_b64 %2, %16 offset:8\n\t"
                 "ds_read_b64 %3, %15 offset:8\n\t"
                 "ds_read_b64 %4, %16\n\t"
                 "ds_read_b64 %5, %14 offset:8\n\t"
                 "ds_read_b64 %6, %17\n\t"
                 "ds_read_b64 %7, %18 offset:64\n\t"
                 "ds_read_b64 %8, %18 offset:72\n\t"
                 "ds_read_b64 %9, %18 offset:80\n\t"
                 "ds_read_b64 %10, %18 offset:88\n\t"
                 "ds_read_b64 %11, %18 offset:96\n\t"
                 "ds_read_b64 %12, %18 offset:104\n\t"
                 "ds_read_b64 %13, %18 offset:112\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(w[0]), "=&v"(w[1]),
                   "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6])
                 : "v"(a3), "v"(a4), "v"(pd), "v"(a6), "v"(aw));
#pragma unroll
    for (int d = 0; d < 7; ++d) acc = __builtin_fma(w[d], v[d], acc);
    return acc;
}

__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// NS: streams the loaders read per slot -- 1: x;  2: x and x2 (p-update);  3: x, x2 and xacc or x3 (pending x-updates)
// SRC / OUT: the launch has a source vector / stores its result (a dead CG step stores nothing)
// SLOT: the interior entries of a slab's evaluation list carry their output slots (the restriction through the window: the even
// nodes of the lattice, output = their COARSE slots); otherwise they are one run of consecutive slots
template <bool FUSED, int NS, bool SRC, bool OUT, int NLW, bool SLOT = false>
__global__ void __launch_bounds__(S2_NT, 4)
k_apply_slab2(LevelDev lv, const double *__restrict__ coef, const uint16_t *__restrict__ dmask, ApplyArgs a, SlabTables st)
{
    constexpr int NDIR = 15, NTERM = 7;
    constexpr int NL = 64 * NLW, NE = S2_NT - NL, S2_SMAX = s2_smax(NLW);
    constexpr int S2_U = s2_umax(NLW);
    extern __shared__ double smem[];
    double *Wt = smem;                              // [2][S2_WSZ]  class weight tables of the cell being evaluated / being loaded
    double *red = smem + 2 * S2_WSZ;                // [2][S2_RED]
    int *tab = (int *)(red + 2 * S2_RED);           // [S2_MAXSLAB][16] per-slab constants (below)
    int *cells = tab + 2 * S2_TAB;                  // [S2_MAXCELLS] this workgroup's cells (no global load, hence no wait, at a cell change)
    double *img0 = red + 2 * S2_RED + S2_TAB + S2_MAXCELLS / 2 + S2_GUARD;   // [2][win]     the two windows, S2_GUARD zeros below
    const int win = st.lds_nodes;
    const int tid = threadIdx.x;
    const int wave = sgpr(tid >> 6);
    const bool loader = wave < NL / 64;
    const int rt = loader ? tid : tid - NL;         // thread index inside the role
    const int m = lv.m, nsl = st.nslab;
    const int64_t G = gridDim.x, b0 = blockIdx.x;
    const int64_t ncl = (a.nwork - b0 + G - 1) / G;                 // cells of this workgroup: b0, b0 + G, ..
    const int64_t T = ncl * nsl;                                    // steps: (cell, slab) pairs
    if (tid < S2_GUARD) img0[tid - S2_GUARD] = 0.0;
    for (int64_t ci = tid; ci < ncl; ci += S2_NT) {
        const int64_t idx = b0 + ci * G;
        cells[ci] = a.cell_list ? a.cell_list[idx] : (int)idx;
    }
    auto cell_at = [&](int64_t ci) { return (int64_t)sgpr(cells[ci]); };
    // per-slab constants, formed once:  0 ld_off  1 ld_cnt  2 rows of loads  3 lo  4 halo count  5 halo source offset  6 guard start
    //                                   8 cp_off  9 cp_surf  10 n_int  11 first interior slot  12 interior rows  13 surface rows
    if (tid < nsl) {
        auto plane_off = [&](int k) {   // PO(k) = number of lattice nodes in planes < k
            if (k <= 0) return 0;
            if (k > m + 1) k = m + 1;
            const long long n1 = m + 1, n2 = m + 1 - k;
            return (int)((n1 * (n1 + 1) * (n1 + 2) - n2 * (n2 + 1) * (n2 + 2)) / 6);
        };
        const int *hd = st.head + 8 * tid;
        const int k0 = hd[0], k1 = tid + 1 < nsl ? hd[8] : m + 1, kp = tid > 0 ? hd[-8] : 0;
        const int lo = plane_off(k0 - 1), lo_prev = plane_off(kp - 1);
        int *t = tab + 16 * tid;
        t[0] = hd[1];
        t[1] = hd[2];
        t[2] = (hd[2] + s2_rows(NS, NLW) * NL - 1) / (s2_rows(NS, NLW) * NL);
        t[3] = lo;
        t[4] = tid > 0 ? plane_off(k0 + 1) - lo : 0;
        t[5] = lo - lo_prev;
        t[6] = plane_off(k1 + 1) - lo;
        const int cp_off = hd[3], cp_cnt = hd[4], cp_surf = hd[5], n_int = cp_cnt - cp_surf;
        t[8] = cp_off;
        t[9] = cp_surf;
        t[10] = n_int;
        t[11] = n_int > 0 ? (int)st.cp_slot[cp_off + cp_surf] : 0;     // (interior slots of a slab are consecutive: host check)
        t[12] = (n_int + NE - 1) / NE;
        t[13] = (cp_surf + NE - 1) / NE;
    }
    __syncthreads();

    if (loader) {
        // ------------------------------------------------------------------------------------------------------------------
        // LOADERS.  Rows of 512 slots: row j of step f = (cell, slab) holds entries (j R + e) 256 + rt of the slab's load list; the
        // rows of all steps of all this workgroup's cells are ONE stream, walked by three cursors a fixed distance apart.
        // ------------------------------------------------------------------------------------------------------------------
        const double beta = to_sgpr(FUSED && NS >= 2 ? a.scal[a.s_num] / a.scal[a.s_den] : 0.0);
        const double ax = to_sgpr(FUSED && NS == 3 ? a.scal[a.a_num] / a.scal[a.a_den] : 0.0);
        const double c2 = to_sgpr(FUSED && a.x3 ? a.scal[a.c_num] / a.scal[a.c_den] : 0.0);
        const bool x3mode = FUSED && NS == 3 && a.x3 != nullptr;
        const bool xzero = FUSED && (a.flags & 128);
        const double *xa_base = NS == 3 ? (a.x3 ? a.x3 : a.xacc) : nullptr;
        struct Cursor {
            int64_t ci;
            int sl, j, nrow, ld_off, ld_cnt;
            int64_t cell;
            bool live;      // false: behind the last row (stays there: its loads repeat the last row's)
        };
        auto enter_slab = [&](Cursor &c) {
            const int *t = tab + 16 * c.sl;
            c.ld_off = sgpr(t[0]);
            c.ld_cnt = sgpr(t[1]);
            c.nrow = sgpr(t[2]);
        };
        auto advance = [&](Cursor &c) {
            if (!c.live) return;
            if (++c.j < c.nrow) return;
            if (c.sl + 1 == nsl && c.ci + 1 == ncl) {      // the last row of the stream: stay
                --c.j;
                c.live = false;
                return;
            }
            c.j = 0;
            if (++c.sl == nsl) {
                c.sl = 0;
                ++c.ci;
                c.cell = cell_at(c.ci);
            }
            enter_slab(c);
        };
        Cursor cw{0, 0, 0, 0, 0, 0, cell_at(0), true};
        enter_slab(cw);
        Cursor cd = cw, cp = cw;
        constexpr int D = s2_depth(NS, NLW), R = s2_rows(NS, NLW);
        uint32_t wr[2 * D][R];                      // addressing words: row i -> wr[i % (2 D)]
        double dx[D][R], d2[D][R], da[D][R];        // data: row i -> d*[i % D]
        auto load_words = [&](const Cursor &c, uint32_t (&w)[R]) {
#pragma unroll
            for (int e = 0; e < R; ++e) {
                const int v = min((c.j * R + e) * NL + rt, c.ld_cnt - 1);     // (rows are completed with copies of their last entry)
                w[e] = st.ld_word[c.ld_off + v];
            }
        };
        auto load_data = [&](const Cursor &c, const uint32_t (&w)[R], double (&x)[R], double (&x2)[R], double (&xa)[R]) {
            const double *bx = a.x + c.cell * lv.ld, *b2 = NS >= 2 ? a.x2 + c.cell * lv.ld : nullptr,
                         *ba = NS == 3 ? xa_base + c.cell * lv.ld : nullptr;
#pragma unroll
            for (int e = 0; e < R; ++e) {
                const uint32_t t = w[e] >> 16;
                x[e] = bx[t];
                if (NS >= 2) x2[e] = b2[t];
                if (NS == 3) xa[e] = ba[t];
            }
        };
#pragma unroll
        for (int u = 0; u < 2 * D; ++u) {
            load_words(cw, wr[u]);
            advance(cw);
        }
#pragma unroll
        for (int u = 0; u < D; ++u) {
            load_data(cd, wr[u], dx[u], d2[u], da[u]);
            advance(cd);
        }
        double rr = 0.0;
        int64_t rows_left = 0;                      // rows of the whole stream
        for (int s = 0; s < nsl; ++s) rows_left += sgpr(tab[16 * s + 2]);
        rows_left *= ncl;
        S2_T0();
        for (; rows_left > 0; rows_left -= 2 * D) {
#pragma unroll
            for (int u = 0; u < 2 * D; ++u) {
                const bool row_live = u < rows_left;                // (the stream's length need not be a multiple of 2 D)
                if (row_live) {
                    const int nb = (int)((cp.ci * nsl + cp.sl) & 1);
                    double *dst = img0 + (size_t)nb * win;
                    const int *t = tab + 16 * cp.sl;
                    const int lo = sgpr(t[3]);
                    if (cp.j == 0) {
                        // ---- a new window ----
                        if (cp.sl == 0) {
                            // class weight table of this cell for its evaluators: W[class][dir] = sum_t ctab[class][dir][t] * s[t]
                            double *W = Wt + (size_t)(cp.ci & 1) * S2_WSZ;
                            if (rt < lv.ncls * NDIR) {
                                const int cls = rt / NDIR, d = rt - cls * NDIR;
                                double w = 0.0;
                                if (a.flags & 4) {   // cell-independent stencil (restriction): the weight is the last term of the class table
                                    w = lv.ctab[(size_t)rt * NTERM + NTERM - 1];
                                } else {
                                    double s[NTERM];
                                    cell_scales<3>(coef + cp.cell * 8, a.alpha, a.lambda, s, a.flags);
                                    const double *c = lv.ctab + (size_t)rt * NTERM;
#pragma unroll
                                    for (int q = 0; q < NTERM; ++q) w += c[q] * s[q];
                                }
                                W[cls * S2_WROW + d] = w;
                            } else if (rt < lv.ncls * NDIR + lv.ncls) {
                                W[(rt - lv.ncls * NDIR) * S2_WROW + 15] = 0.0;        // (row padding: read, never used)
                            }
                        } else {
                            // planes k0-1 and k0 (last evaluated plane and upper halo of the previous slab): from the other window
                            const double *old = img0 + (size_t)(nb ^ 1) * win;
                            const int cnt = sgpr(t[4]), src = sgpr(t[5]);
                            for (int q = rt; q < cnt; q += NL) dst[q] = old[src + q];
                        }
                        for (int q = sgpr(t[6]) + rt; q < win; q += NL) dst[q] = 0.0;        // zero guard behind the last plane
                    }
                    // ---- the row: every slot once; all of a slot's loads have landed before its stores (xout / xacc may alias x2) ----
                    double *bxa = FUSED && NS == 3 && !x3mode ? a.xacc + cp.cell * lv.ld : nullptr;
                    double *bxo = FUSED && a.xout ? a.xout + cp.cell * lv.ld : nullptr;
#pragma unroll
                    for (int e = 0; e < R; ++e) {
                        const uint32_t w = wr[u][e];
                        const bool valid = (cp.j * R + e) * NL + rt < cp.ld_cnt;
                        const uint32_t slot = w >> 16;
                        double val = xzero ? 0.0 : dx[u % D][e];
                        if (FUSED) {
                            const double x2v = NS >= 2 ? d2[u % D][e] : 0.0, xav = NS == 3 ? da[u % D][e] : 0.0;
                            if (NS == 3 && !x3mode) {
                                if (valid) st_global(bxa + slot, axpy1(ax, x2v, xav));
                            }
                            if (NS == 3 && x3mode) {
                                const double t1 = axpy1(ax, x2v, val);
                                const double p2 = axpy1(beta, x2v, xav);
                                val = axpy1(c2, p2, t1);
                            } else if (NS >= 2)
                                val = axpy1(beta, x2v, val);
                            if (bxo) {
                                if (valid) st_global(bxo + slot, val);
                            }
                            if (valid) rr += val * val;
                        }
                        if (valid) dst[(int)(w & 0xffffu) - lo] = val;
                    }
                    if (cp.j == cp.nrow - 1) {
                        // ---- the window is full ----
                        if (FUSED && cp.sl == nsl - 1) {                       // the cell's r.r: one partial per loader wave
                            const double s = wave_sum63(rr);
                            if ((tid & 63) == 63) red[(cp.ci & 1) * S2_RED + (16 - NLW) + wave] = s;
                            rr = 0.0;
                        }
                        S2_PHASE(0);                                            // (0: filling a window)
                        S2_ARRIVE();
                        __syncthreads();
                        S2_LEAVE();
                        S2_PHASE(1);                                            // (1: waiting for the evaluators)
                    }
                    advance(cp);
                }
                // the row D ahead: its data, with the words requested D rows ago; the row 2 D ahead: its words
                load_data(cd, wr[(u + D) % (2 * D)], dx[u % D], d2[u % D], da[u % D]);
                advance(cd);
                load_words(cw, wr[u]);
                advance(cw);
            }
        }
        __syncthreads();                            // (the evaluators' last step)
        S2_STORE(0);
        S2_STORE_WAVE();
    } else {
        // ------------------------------------------------------------------------------------------------------------------
        // EVALUATORS: at iteration g they evaluate step g from window g & 1
        // ------------------------------------------------------------------------------------------------------------------
        const uint32_t img_a0 = lds_addr(img0);
        const int wbase = (wave - NLW) * 64;        // first entry of this wave in a row of NE
        double pap = 0.0;
        int64_t ci = 0;
        int sl = 0;
        int64_t cell = cell_at(0), cell_next = ncl > 1 ? cell_at(1) : cell;
        uint32_t dm = 0, mq[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
        double w0[NDIR];
        // requested one step ahead, in front of the barrier (nothing of it depends on the windows): addressing words, slots and source
        // values of this thread's surface entries and of its interior entries (all S2_U rows of them: the host sends slabs with
        // more to k_apply_slab).  Rows are completed with copies of their last entry.
        uint32_t sw[S2_SMAX], iw[S2_U];
        int ss[S2_SMAX], is[S2_U];
        double ssv[S2_SMAX], isv[S2_U];
        uint32_t dm_n = 0, mq_n[4];
        // (absent tables: any readable address -- the values are not used then)
        const uint16_t *dmask_p = (a.flags & 1) ? dmask : reinterpret_cast<const uint16_t *>(st.head);
        const uint32_t *mult_p = FUSED && a.mult ? reinterpret_cast<const uint32_t *>(a.mult) : reinterpret_cast<const uint32_t *>(st.head);
        const bool has_dm = (a.flags & 1) != 0, has_mult = FUSED && a.mult != nullptr;
        auto request = [&](int sl_n, int64_t cell_n) {
            dm_n = (uint32_t)dmask_p[has_dm ? cell_n : 0];
#pragma unroll
            for (int q = 0; q < 4; ++q) mq_n[q] = mult_p[has_mult ? cell_n * 4 + q : q];
            const int *t = tab + 16 * sl_n;
            const int cp_off = sgpr(t[8]), cp_surf = sgpr(t[9]), n_int = sgpr(t[10]), t_first = sgpr(t[11]);
            const double *sc = SRC ? a.src + cell_n * lv.ld : nullptr;
#pragma unroll
            for (int q = 0; q < S2_SMAX; ++q) {
                const int v = max(min(q * NE + rt, cp_surf - 1), 0);      // (a slab without surface entries: entry 0, unused)
                sw[q] = st.cp_word[cp_off + v];
                ss[q] = (int)st.cp_slot[cp_off + v];
            }
#pragma unroll
            for (int q = 0; q < S2_U; ++q) {
                const int v = max(min(q * NE + rt, n_int - 1), 0);
                iw[q] = st.cp_word[cp_off + cp_surf + v];
                if (SLOT) is[q] = (int)st.cp_slot[cp_off + cp_surf + v];
                if (SRC) isv[q] = sc[t_first + v];
            }
            if (SRC) {
#pragma unroll
                for (int q = 0; q < S2_SMAX; ++q) ssv[q] = sc[ss[q]];
            }
        };
        // The results of step g are STORED AT THE START OF STEP g + 1, behind the wait for that step's requested values.  The memory
        // counter of a wave counts loads and stores alike and in order: a wait for requested values is also a wait for every store
        // issued before it, and a store is slow to issue and to be acknowledged while the loaders keep the CU's memory pipeline full
        // (per-wave step times, round 5: 5.2 us per step with the evaluators' stores, 3.9 us with none).  Stored one step late, the
        // youngest store a wait meets is a whole step old, and the step's evaluation holds no memory instruction at all.
        // (Same-box A/B on config 5's share: 6.60-6.62 -> 6.33-6.37 ms per apply on one box, within the noise on two others.)
        double od[S2_U], sd[S2_SMAX];               // results of the previous step: interior rows, surface rows
        int dis[S2_U], dss[S2_SMAX];                // their output slots (interior: SLOT launches only)
        double *d_oc = nullptr;
        int d_nint = 0, d_tfirst = 0, d_surf = 0;
        auto flush = [&]() {
            if (!OUT) return;
            if constexpr (SLOT) {
#pragma unroll
                for (int q = 0; q < S2_U; ++q) {
                    const int v = q * NE + rt;
                    if (q * NE + wbase < d_nint) {
                        if (v < d_nint) st_global(d_oc + dis[q], od[q]);
                    }
                }
            } else {
                // interior rows in pairs, 16 bytes per lane: the even lane of a lane pair stores row q's two results (its own and
                // its neighbour's), the odd lane row q + 1's -- half the store instructions (their issue is what an evaluator waits for)
                const int odd = rt & 1;
#pragma unroll
                for (int q = 0; q < S2_U; q += 2) {
                    if (q * NE + wbase < d_nint) {
                        const double other = pair_swap_f64(odd ? od[q] : od[q + 1]);
                        const double lo = odd ? other : od[q], hi = odd ? od[q + 1] : other;
                        const int v0 = (q + odd) * NE + (rt & ~1);              // the pair's first entry
                        double *dst = d_oc + (d_tfirst + v0);
                        if (v0 + 1 < d_nint)
                            st_global2(dst, lo, hi);
                        else if (v0 < d_nint)
                            st_global(dst, lo);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < S2_SMAX; ++q) {
                const int v = q * NE + rt;
                if (q * NE + wbase < d_surf) {
                    if (v < d_surf) st_global(d_oc + dss[q], sd[q]);
                }
            }
        };
        request(0, cell);
        S2_CYC_DECL();
        S2_T0();
        __syncthreads();                                            // (the loaders fill window 0)
        S2_LEAVE();
        S2_PHASE(3);
        for (int64_t g = 0; g < T; ++g) {
            const int b = (int)(g & 1);
            {
                // (requested with the step: the same values at every slab of a cell)
                dm = has_dm ? (uint32_t)sgpr((int)dm_n) : 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) mq[q] = has_mult ? (uint32_t)sgpr((int)mq_n[q]) : 0x01010101u;
            }
            // (the requested values have arrived: everything below this statement -- the stores of the previous step's results first --
            //  is issued behind the wait for them)
#pragma unroll
            for (int q = 0; q < S2_U; ++q) {
                asm volatile("" : "+v"(iw[q]));
                if (SLOT) asm volatile("" : "+v"(is[q]));
                if (SRC) asm volatile("" : "+v"(isv[q]));
            }
#pragma unroll
            for (int q = 0; q < S2_SMAX; ++q) {
                asm volatile("" : "+v"(sw[q]), "+v"(ss[q]));
                if (SRC) asm volatile("" : "+v"(ssv[q]));
            }
            S2_PHASE(4);                                            // (4: waiting for the requested values)
            flush();
            S2_PHASE(5);                                            // (5: issuing the previous step's stores)
            if (sl == 0) {
                const double *W = Wt + (size_t)(ci & 1) * S2_WSZ;
#pragma unroll
                for (int d = 0; d < NDIR; ++d) w0[d] = to_sgpr(lds_ld(W + d));   // interior weight row (class 0), SGPR-resident
            }
            const int *t = tab + 16 * sl;
            const int cp_surf = sgpr(t[9]), n_int = sgpr(t[10]), t_first = sgpr(t[11]), lo = sgpr(t[3]);
            const uint32_t img_a = img_a0 + (uint32_t)b * 8u * (uint32_t)win;     // LDS byte address of lattice position lo
            const uint32_t w_a = lds_addr(Wt + (size_t)(ci & 1) * S2_WSZ);
            d_oc = OUT ? a.out + cell * (a.out_ld ? a.out_ld : (int64_t)lv.ld) : nullptr;
            d_nint = n_int;
            d_tfirst = t_first;
            d_surf = cp_surf;

            // cell interior: one weight row for all nodes
#pragma unroll
            for (int q = 0; q < S2_U; ++q) {
                const int v = q * NE + rt;
                if (q * NE + wbase < n_int) {                                   // (this wave's 64 entries of the row: none beyond the list)
                    double x[15];
                    int L, len, A, B;
                    S2_CYC(0);
                    decode_lattice(iw[q], m, L, len, A, B);
                    const uint32_t ap = img_a + 8u * (uint32_t)(L - lo);
                    S2_CYC(1);
                    read15(ap, len, A, B, x);
                    S2_CYC(2);
                    double acc = w0[1] * x[1];                                  // (the order hipcc contracts stencil_eval_v to)
                    acc = __builtin_fma(w0[0], x[0], acc);
#pragma unroll
                    for (int d = 2; d < NDIR; ++d) acc = __builtin_fma(w0[d], x[d], acc);
                    double o = (SRC ? isv[q] : 0.0) + acc;
                    S2_CYC_ADD(o);
                    od[q] = o;
                    if (SLOT) dis[q] = is[q];
                    if (FUSED && v < n_int) pap = __builtin_fma(x[0], o, pap);
                }
            }
            S2_PHASE(0);                                            // (0: cell interior)
            // surface entities: the class weight row comes from the LDS table
#pragma unroll
            for (int q = 0; q < S2_SMAX; ++q) {
                if (q * NE + wbase < cp_surf) {                       // (a wave none of whose 64 entries is in the list skips the row)
                    const int v = q * NE + rt;
                    int L, len, A, B, cls, k;
                    decode32w(sw[q], m, L, len, A, B, cls, k);
                    double ctr;
                    double o = surface_eval(img_a + 8u * (uint32_t)(L - lo), w_a + 8u * (uint32_t)(cls * S2_WROW), len, A, k ? B : 0, ctr);
                    o = (SRC ? ssv[q] : 0.0) + o;
                    if ((dm >> (cls - 1)) & 1u) o = 0.0;
                    sd[q] = o;
                    dss[q] = ss[q];
                    if (FUSED && v < cp_surf) {
                        const int en = cls - 1;
                        const uint32_t word = en < 4 ? mq[0] : en < 8 ? mq[1] : en < 12 ? mq[2] : mq[3];
                        const uint32_t mu = (word >> (8 * (en & 3))) & 0xffu;
                        pap += (double)mu * (ctr * o);
                    }
                }
            }
            const bool cell_done = sl == nsl - 1;
            const int par = (int)(ci & 1);
            const int64_t cell_was = cell;
            if (FUSED && cell_done) {                               // the cell's p.Ap: one partial per evaluator wave
                const double s = wave_sum63(pap);
                if ((tid & 63) == 63) red[par * S2_RED + (wave - NL / 64)] = s;
                pap = 0.0;
            }
            if (++sl == nsl) {
                sl = 0;
                ++ci;
                cell = cell_next;
                cell_next = ci + 1 < ncl ? cell_at(ci + 1) : cell;
            }
            if (g + 1 < T) request(sl, cell);
            S2_PHASE(1);                                            // (1: surface entities)
            S2_ARRIVE();
            __syncthreads();
            S2_LEAVE();
            S2_PHASE(2);                                            // (2: waiting for the loaders)
            if (FUSED && cell_done && rt == 0) {                    // (partials of this parity are written again two cells on)
                double s_pap = 0.0, s_rr = 0.0;
#pragma unroll
                for (int q = 0; q < 16 - NLW; ++q) s_pap += red[par * S2_RED + q];
#pragma unroll
                for (int q = 0; q < NLW; ++q) s_rr += red[par * S2_RED + (16 - NLW) + q];
                a.blockpart[2 * cell_was] = s_pap;
                a.blockpart[2 * cell_was + 1] = s_rr;
            }
        }
        flush();                                                    // (the last step's results)
        S2_STORE(8);
        S2_STORE_WAVE();
        S2_CYC_STORE();
    }
}

size_t slab2_lds_bytes(const MeshDev &mesh)
{
    return sizeof(double) * (size_t)(2 * S2_WSZ + 2 * S2_RED + S2_TAB + S2_MAXCELLS / 2 + S2_GUARD + 2 * (size_t)mesh.slab.lds_nodes);
}

}  // namespace

bool apply_slab2_ok(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a)
{
    // (flags bit 2, the restriction through the window, keeps k_apply_slab: its evaluation lists do not hold consecutive slots;
    //  bit 3, the driver integrals, has its own instantiations there)
    const int nlw = (a.flags & 4) ? 8 : 4;      // loader waves (launch_slab2)
    if (!L.apply_slab2 || lv.dim != 3 || lv.ncls != 15 || lv.m + 2 > S2_GUARD || !mesh.slab.head || mesh.slab.nslab < 2 || mesh.slab.nslab > S2_MAXSLAB ||
        mesh.slab.max_surf > s2_smax(nlw) * (S2_NT - 64 * nlw) || mesh.slab.max_int > s2_umax(nlw) * (S2_NT - 64 * nlw) || (a.flags & 8) || (((a.flags & 4) != 0) != (a.out_ld != 0)) ||
        a.xcoarse || a.rcoarse)      // (level transfers folded into an apply: the LDS-resident kernels of levels 5 and 6 only)
        return false;
    return slab2_lds_bytes(mesh) <= 160 * 1024;
}

template <bool FUSED, int NS, bool SRC, bool OUT, bool SLOT = false>
static void launch_slab2(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a)
{
    auto kern = k_apply_slab2<FUSED, NS, SRC, OUT, SLOT ? 8 : 4, SLOT>;
    const size_t bytes = slab2_lds_bytes(mesh);
    HMG_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const int64_t nblocks = a.cell_list ? a.ncell_list : a.ncells_prefix ? a.ncells_prefix : mesh.ncells;
    if (nblocks == 0) return;
    ApplyArgs b = a;
    b.nwork = nblocks;
    if (L.cell_order && !a.cell_list && !a.ncells_prefix && mesh.cell_perm) {   // (XCD-aware cell order: workgroup b -> XCD b % 8, and
        b.cell_list = mesh.cell_perm;                                           //  the grid is a multiple of 8, so b + i G stays there)
        b.ncell_list = nblocks;
    }
    // one resident workgroup per CU walks its cells (option slab2_grid: another number -- a multiple of 8 keeps a workgroup's cells on its XCD)
    const int64_t grid = std::min<int64_t>(nblocks, L.slab2_grid > 0 ? L.slab2_grid : (int64_t)L.num_cu);
    if ((nblocks + grid - 1) / grid > S2_MAXCELLS) throw std::runtime_error("operator apply: more cells per workgroup than the slab kernel stages");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(S2_NT), bytes, L.stream, lv, mesh.coef, mesh.dmask, b, mesh.slab);
    check_launch();
}

void launch_apply_slab2(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a, bool fused)
{
    if (!a.x) throw std::runtime_error("operator apply: null input vector");
    if (!fused && !a.out) throw std::runtime_error("operator apply: a plain launch needs an output vector");
    if (fused && (!a.blockpart || !a.scal)) throw std::runtime_error("operator apply: fused launch without its reduction scratch");
    if (!mesh.coef && !(a.flags & 4)) throw std::runtime_error("operator apply: no operator coefficients on the device (hmg_grid_set_operator)");
    if ((a.flags & 1) && !mesh.dmask) throw std::runtime_error("operator apply: constraint requested without a Dirichlet mask");
    if (a.xcoarse || a.rcoarse) throw std::runtime_error("operator apply: the slab kernel folds no level transfer");
    if (fused && (a.xacc || a.x3) && !a.x2) throw std::runtime_error("operator apply: a pending x-update without its direction vector");
    if (fused && a.xacc && a.x3) throw std::runtime_error("operator apply: xacc and x3 exclude each other");
    if ((a.flags & 128) && !(fused && a.x3 && a.x2 && a.xout))
        throw std::runtime_error("operator apply: the zero-input form exists for the residual with two pending x-updates only");
    if (L.n_slab2_launches) *L.n_slab2_launches += 1;
    const bool src = a.src != nullptr, out = a.out != nullptr;
    if (a.flags & 4) {   // the restriction through the window (launch_restrict_slab)
        if (fused || src || !out || !a.out_ld) throw std::runtime_error("restriction through the window: plain launch with its own column stride");
        launch_slab2<false, 1, false, true, true>(L, lv, mesh, a);
        return;
    }
    if (!fused) {
        if (src)
            launch_slab2<false, 1, true, true>(L, lv, mesh, a);
        else
            launch_slab2<false, 1, false, true>(L, lv, mesh, a);
        return;
    }
    const int ns = (a.xacc || a.x3) ? 3 : a.x2 ? 2 : 1;
#define S2_CASE(NSV, SRCV, OUTV) \
    if (ns == NSV && src == SRCV && out == OUTV) return launch_slab2<true, NSV, SRCV, OUTV>(L, lv, mesh, a)
    S2_CASE(1, false, true);
    S2_CASE(1, false, false);
    S2_CASE(2, false, true);
    S2_CASE(2, false, false);
    S2_CASE(3, false, true);
    S2_CASE(3, false, false);
    S2_CASE(1, true, true);
    S2_CASE(2, true, true);
    S2_CASE(3, true, true);
    S2_CASE(1, true, false);
    S2_CASE(2, true, false);
    S2_CASE(3, true, false);
#undef S2_CASE
}

}  // namespace hmg
