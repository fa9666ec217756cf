// HIP kernels for gfx950 (MI355X): per-cell lattice-stencil operator apply, interface sum,
// Dirichlet / duplicate masks, level transfer, fused CG vector kernels, level-1 gather/scatter and
// a (Chebyshev-)preconditioned CG for the coarse system.  Wave = 64 lanes everywhere.
//
// Reference behaviour reproduced (file:line in the reference checkout):
//   k_apply, k_apply_slab  src/apply_local_operators.jl:85-133 (+ :7-27 residual, + constraint mask,
//                          + the p-update and dot products of src/multigrid.jl:54-68 when FUSED)
//   k_iface_*           src/implicit_fine_grid.jl:209-328   (copies summed in ascending cell order)
//   k_mask              src/implicit_fine_grid.jl:94-139 / :334-386
//   k_restrict/prolong  src/interpolation.jl:52-74
//   k_cg_*, k_dot*      src/multigrid.jl:46-71
//   k_gather/scatter    src/implicit_fine_grid.jl:148-202
#include "hmg_device.hpp"
#include "hmg_stencil.hpp"

#include <cstdio>
#include <stdexcept>
#include <string>

namespace hmg {

__global__ void __launch_bounds__(256) k_finalize(const double *__restrict__ partials, int n, double *scal, int slot)
{
    __shared__ double red[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partials[i];
    double s = block_sum(v, red);
    if (threadIdx.x == 0) scal[slot] = s;
}

// ---------------------------------------------------------------------------------------------
// operator apply
// ---------------------------------------------------------------------------------------------
#ifdef HMG_PHASE_TIMING   // dev build (make phase-timing): thread 0 of every workgroup stamps its phases into blockpart
#define HMG_STAMP(i)                                                                     \
    do {   /* stamps behind the 2 * ncells reduction partials (full-grid launches only) */ \
        if (a.blockpart && !a.cell_list && tid == 0)                                     \
            a.blockpart[2 * (size_t)a.nwork + 8 * cell + (i)] = (double)wall_clock64(); \
    } while (0)
// k_apply_slab: thread 0 adds up what it spends per phase over the cell's slabs (its own time line, barrier waits included)
#define HMG_SLAB_T0() const long long _t0 = wall_clock64(); long long _tl = _t0; double _ph[5] = {0, 0, 0, 0, 0}
#define HMG_SLAB_PHASE(i)                              \
    do {                                               \
        const long long _n = wall_clock64();           \
        _ph[i] += (double)(_n - _tl);                  \
        _tl = _n;                                      \
    } while (0)
#define HMG_SLAB_STORE()                                                                      \
    do {                                                                                      \
        if (a.blockpart && tid == 0) {                                                        \
            double *_o = a.blockpart + 2 * (size_t)a.nwork + 8 * cell;                        \
            _o[0] = (double)_t0;                                                              \
            for (int _q = 0; _q < 5; ++_q) _o[1 + _q] = _ph[_q];                              \
            _o[6] = (double)wall_clock64();                                                   \
        }                                                                                     \
    } while (0)
#else
#define HMG_STAMP(i)
#define HMG_SLAB_T0()
#define HMG_SLAB_PHASE(i)
#define HMG_SLAB_STORE()
#endif

// WD: instantiation for the driver integrals (flags bit 3); CG: the folded prolongation stages the coarse column at the even nodes
// of the lattice image itself (flags bit 6; cells that fill a third of the LDS have no room for it behind the image)
// RS: the results are restricted to the coarser level in the epilogue (ApplyArgs::rcoarse, see the end of the kernel)
// WC (register-blocked instantiations): the class weight rows come from the class-weight cache of hmg_grid_set_operator
// (LevelDev::wcache, one row set per distinct coefficient row and sign of alpha: hmg_apply_wave.hip) instead of being combined
// from the class table and the cell's seven scales by every wave: ONE load per lane, requested in front of the column loads, no
// table terms held across the load phase, no products -- the same weights to the last bit (same products, same order, formed once)
// LF (fused WC instantiations): the form of the load phase is known at compile time -- 1: one stream in one batch (CG step 0),
// 2: two streams in one batch, nothing stored (a dead step), 3: the batched general form; 0: decided at run time.  One form per
// instantiation: the other forms' values do not compete for the 80 registers, and the dead step can take its rows behind the
// column loads (two whole columns in one batch are 65 registers in flight: with the rows in front it lost 0.5 ms) while the
// others take them in front.
template <int DIM, int NT, int SPT, bool FUSED, int RB, bool WD = false, bool CG = false, bool RS = false, bool WC = false, int LF = 0>
__global__ void __launch_bounds__(NT, NT >= 640 ? 8 : RB && NT == 512 ? 6 : 1)   // 2 x 1024 threads per CU need <= 64 VGPRs, 3 x 512: 80
k_apply(LevelDev lv, const double *__restrict__ coef, const uint16_t *__restrict__ dmask, ApplyArgs a)
{
    constexpr int NDIR = DIM == 3 ? 15 : 7;
    constexpr int NTERM = DIM == 3 ? 7 : 4;
    constexpr int WS = RB ? WSZ_RB : WSZ;
    extern __shared__ double smem[];
    double *W = smem;
    double *xs = smem + WS + lv.lds_g0;
    const int tid = threadIdx.x;
    // One-wave workgroups (cells of up to 192 nodes) are PERSISTENT: a launch of 196 608 of them is bound by the rate at
    // which waves can be started and by what every start repeats (kernel arguments, LDS allocation), not by the cells'
    // work -- so the launcher starts as many as are resident at once and each walks its share of the cells.  (The body
    // below is not indented for the loop: every other instantiation runs it exactly once.)
    // (round 4: a persistent loop around the one-form level-6 instantiations LF = 1 / 2 -- 70 VGPRs without it -- still spills 552 /
    //  672 B per lane: the backend hoists the addresses of every table and column load out of the loop)
    constexpr bool LOOP = NT == 64;
    int64_t blk = blockIdx.x;
    do {
    const int64_t cell = a.cell_list ? (int64_t)a.cell_list[blk] : blk;
    const int nf = lv.nf;
    HMG_STAMP(0);

    // Class weight table of this cell: W[class][dir] = sum_t ctab[class][dir][t] * s[t].  Done by the LAST
    // waves of the workgroup (their share of the column load below is the smallest), while the others are
    // already issuing their column loads -- the table's L2 latency then overlaps the HBM latency.
    // RB instantiations share no weight table: every wave combines the class rows it needs itself (below, behind its
    // column loads); the cell's coefficients are requested first, one per lane
    size_t wbase16 = 0;                 // WC: first entry of this cell's row set in the cache (wave-uniform: scalar loads)
    if constexpr (WC) wbase16 = (size_t)(2 * HMG_KP(int32_t, a.cell_class)[cell] + (a.alpha < 0.0 ? 1 : 0)) * WAVE_WSTRIDE;
    const double cv = RB && !WC ? coef[cell * 8 + (tid & 7)] : 0.0;
    if (!RB && (tid >= NT - 256 || NT <= 256)) {
        const int first = NT <= 256 ? tid : tid - (NT - 256);
        const int step = NT <= 256 ? NT : 256;
        if constexpr (WC) {
            // the whole class table of this cell's class from the cache: 225 loads per cell instead of 1575 table terms and as
            // many products -- for the one-wave workgroups of the small levels the table was most of a cell's work (level 4: 12.6 KB
            // of table terms per 1.3 KB column)
            for (int idx = first; idx < lv.ncls * NDIR; idx += step)
                W[idx] = lv.wcache[wbase16 + (size_t)((idx / NDIR) * WAVE_ROW + idx % NDIR)];
        } else {
        double s[NTERM];
        cell_scales<DIM>(coef + cell * 8, a.alpha, a.lambda, s, a.flags);
        for (int idx = first; idx < lv.ncls * NDIR; idx += step) {
            const double *c = lv.ctab + (size_t)idx * NTERM;
            double w = 0.0;
#pragma unroll
            for (int t = 0; t < NTERM; ++t) w += c[t] * s[t];
            W[idx] = w;
        }
        }
    }
    for (int q = tid; q < lv.lds_g0; q += NT) smem[WS + q] = 0.0;
    for (int q = tid; q < lv.lds_g1; q += NT) xs[nf + q] = 0.0;
    // Roles and table words.  (pos32 / sweep32 / sweep_slot / blk_* carry TABLE_PAD entries of padding -- zero words, slot
    // 0xffff -- so the two-ahead prefetch needs no bounds checks and both loops have a wave-uniform trip count.)
    // RB: one entity class per wave.  NT/64/4 waves share a face (FI runs of 64 slots each), the edges are dealt to the
    // waves from the last one backwards (NE per wave: 1 of 16 waves, 2 of 4), the last wave takes the corners instead
    // of an interior block -- the 945 blocks of level 6 fill waves 0..14, the 152 of level 5 waves 0..2.
    // 512-thread instantiation (three workgroups per CU): 2 waves per face with 4 runs each, the interior blocks in two passes
    const int m = lv.m;
    constexpr int NW = NT / 64, WPF = NW / 4 > 0 ? NW / 4 : 1, NE = NW >= 6 ? 1 : 2;
    constexpr int FI = RB && NT == 512 ? 4 : 2;
    constexpr int NPASS = RB && NT == 512 ? 2 : 1;
    // EARLY (register-blocked instantiations with registers to spare, <= 512 threads): the class rows and the
    // multiplicities are requested BEFORE the column loads and combined while the first batch of them is in flight
    // (vmcnt counts in order: waiting for the rows does not wait for the column) -- two registers live across the load
    // phase -- instead of in an exposed chain of L2 round trips between the load phase and the barrier.  (Fetching the
    // addressing words early as well costs the fused instantiation 60 spilled VGPRs.)
    constexpr bool EARLY = RB && (NT <= 256 || WC);
    const int wave = tid >> 6, lane = tid & 63;
    const int face = wave / WPF, ft0 = (wave % WPF) * (FI * 64);
    int edge[NE], ebase[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        edge[q] = NW - 1 - wave + q * NW;
        if (edge[q] >= lv.nedge) edge[q] = -1;
        ebase[q] = lv.off_edge + (edge[q] < 0 ? 0 : edge[q]) * lv.nei;
    }
    uint32_t fw[FI], ew[NE][1], cw = 0u;
#pragma unroll
    for (int q = 0; q < FI; ++q) fw[q] = 0u;
    uint32_t p0 = 0u, p1 = 0u;
    double wv = 0.0;                 // RB: lane d: interior weight d; lane 16 + d: this wave's face; lane 32 + 16 q + d: its edge q
    double c7[NTERM];
    uint32_t q0 = 0u, q1 = 0u;
    int s0 = 0, s1 = 0;
    uint32_t mq[4] = {0, 0, 0, 0};   // the cell's 16 entity multiplicities, wave-uniform -> SGPRs
    uint32_t mqr[4] = {0, 0, 0, 0};
    uint32_t dm = 0u;
    const uint32_t *iw = RB ? lv.blk_word : lv.sweep32;
    const uint16_t *is = RB ? lv.blk_slot : lv.sweep_slot;
    double cwl = 0.0;                   // WC: the last wave's corner weights (60 of them, one per lane)
    auto issue_rows = [&]() {           // class rows (L2-resident table) x the cell's 7 scales, multiplicities, Dirichlet mask
        if constexpr (WC) {
            // (wave-uniform words by scalar loads: Dirichlet mask, multiplicities; the lane's weight by one vector load)
            if (a.flags & 1) {
                const uint32_t w2 = HMG_KP(uint32_t, dmask)[cell >> 1];
                dm = (cell & 1) ? w2 >> 16 : w2 & 0xffffu;
            }
            if (RB) {
                int row = 0;
                if (lane < 15)
                    row = lane;
                else if (lane >= 16 && lane < 31)
                    row = (1 + face) * WAVE_ROW + lane - 16;
                else if (lane >= 32 && lane < 47 && edge[0] >= 0)
                    row = (1 + lv.nface + edge[0]) * WAVE_ROW + lane - 32;
                else if (NE > 1 && lane >= 48 && lane < 63 && edge[NE - 1] >= 0)
                    row = (1 + lv.nface + edge[NE - 1]) * WAVE_ROW + lane - 48;
                wv = lv.wcache[wbase16 + row];
                if (wave == NW - 1) {
                    const int cl = lane < 60 ? lane : 59;
                    cwl = lv.wcache[wbase16 + (1 + lv.nface + lv.nedge + cl / 15) * WAVE_ROW + cl % 15];
                }
            }
            if (FUSED && a.mult) {
#pragma unroll
                for (int q = 0; q < 4; ++q) mqr[q] = HMG_KP(uint32_t, a.mult)[cell * 4 + q];
            }
            return;
        }
        dm = (a.flags & 1) ? dmask[cell] : 0u;
        if (RB) {
            int row = 0;
            if (lane < 15)
                row = lane;
            else if (lane >= 16 && lane < 31)
                row = (1 + face) * NDIR + lane - 16;
            else if (lane >= 32 && lane < 47 && edge[0] >= 0)
                row = (1 + lv.nface + edge[0]) * NDIR + lane - 32;
            else if (NE > 1 && lane >= 48 && lane < 63 && edge[NE - 1] >= 0)
                row = (1 + lv.nface + edge[NE - 1]) * NDIR + lane - 48;
#pragma unroll
            for (int t = 0; t < NTERM; ++t) c7[t] = lv.ctab[(size_t)row * NTERM + t];
        }
        if (FUSED && a.mult) {
            const uint32_t *mp = reinterpret_cast<const uint32_t *>(a.mult + cell * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) mqr[q] = mp[q];
        }
    };
    auto issue_words = [&]() {          // addressing words of this thread's surface runs and interior blocks
        if (RB) {
#pragma unroll
            for (int q = 0; q < FI; ++q) {
                const int ti = ft0 + q * 64 + lane;
                if (ti < lv.nfi) fw[q] = lv.pos32[lv.off_face + face * lv.nfi + ti];
            }
#pragma unroll
            for (int q = 0; q < NE; ++q) ew[q][0] = edge[q] >= 0 && lane < lv.nei ? lv.pos32[ebase[q] + lane] : 0u;
            if (wave == NW - 1 && lane < lv.ncorner) cw = lv.pos32[lane];
        } else {
            p0 = lv.pos32[tid];
            p1 = lv.pos32[tid + NT];
        }
        q0 = iw[tid];
        q1 = RB && NPASS < 2 ? 0u : iw[tid + NT];
        s0 = (int)is[tid];
        s1 = RB && NPASS < 2 ? 0 : (int)is[tid + NT];
    };
    auto finish_tables = [&]() {
        if constexpr (WC) {
            if (RB && wave == NW - 1 && lane < 60) W[lane] = cwl;      // the corners' rows: the only ones in LDS
            if (FUSED) {
#pragma unroll
                for (int q = 0; q < 4; ++q) mq[q] = a.mult ? mqr[q] : 0x01010101u;
            }
            return;
        }
        if (RB) {
            double sc7[NTERM];
#pragma unroll
            for (int t = 0; t < NTERM - 1; ++t) sc7[t] = (a.flags & 2) ? 0.0 : a.alpha * readlane_f64(cv, t);
            sc7[NTERM - 1] = a.alpha * a.lambda * ((a.flags & 16) ? 1.0 : readlane_f64(cv, NTERM - 1));
#pragma unroll
            for (int t = 0; t < NTERM; ++t) wv += c7[t] * sc7[t];
            if (wave == NW - 1) {        // the corners' rows: per-lane weights, private to this wave (LDS ops of a wave are in order)
                const int cbase = (1 + lv.nface + lv.nedge) * NDIR;        // (the only rows in LDS: stored from W[0] on)
                for (int idx = cbase + lane; idx < lv.ncls * NDIR; idx += 64) {
                    double w = 0.0;
#pragma unroll
                    for (int t = 0; t < NTERM; ++t) w += lv.ctab[(size_t)idx * NTERM + t] * sc7[t];
                    W[idx - cbase] = w;
                }
            }
        }
        if (FUSED) {
#pragma unroll
            for (int q = 0; q < 4; ++q) mq[q] = a.mult ? __builtin_amdgcn_readfirstlane(mqr[q]) : 0x01010101u;   // (no table: all 1)
        }
    };
    // (measured, profiles/r04_experiments.txt: the dead step with its rows in front as well, step 0 with the words of its interior
    //  blocks in front, the dead step through the general instantiation -- no difference beyond +-0.05 ms per launch)
    constexpr bool rows_late = WC && LF == 2;
    // (the one-form instantiations have registers to spare -- 70 of 80 --: their addressing words are requested in front of the
    //  column as well, so that nothing but the barrier stands between the column's arrival and the evaluation)
    constexpr bool WEARLY = false;       // (measured: LF == 1 with its words in front goes from 70 to 80 registers + 16 B of scratch)
    if (EARLY && !rows_late) issue_rows();
    if (WEARLY) issue_words();
    const double *xc = a.x + cell * lv.ld;
    double rr = 0.0, pap = 0.0;
    {
        // Load phase, in batches of HB slots per thread: all loads of a batch first, then its stores / LDS
        // writes, so that a store to xout / xacc (which may alias x2) never sits between two loads of a batch.
        // FUSED extras: xacc += ax * x2 (the previous CG step's x-update, x2 = p_old), xin = x + beta * x2.
        const double *x2c = FUSED && a.x2 ? a.x2 + cell * lv.ld : nullptr;
        double *xoc = FUSED && a.xout ? a.xout + cell * lv.ld : nullptr;
        double *xac = FUSED && a.xacc ? a.xacc + cell * lv.ld : nullptr;
        // x3 mode (two pending CG x-updates folded into a residual): xin = (x + ax*p1) + c2*(r2 + beta*p1), x2 = p1,
        // x3 = r2; the same three roundings as x += ax*p1; p2 = r2 + beta*p1; x += c2*p2 done one after the other
        const double *x3c = FUSED && a.x3 ? a.x3 + cell * lv.ld : nullptr;
        const double beta = x2c ? a.scal[a.s_num] / a.scal[a.s_den] : 0.0;
        const double ax = xac || x3c ? a.scal[a.a_num] / a.scal[a.a_den] : 0.0;
        const double c2 = x3c ? a.scal[a.c_num] / a.scal[a.c_den] : 0.0;
        // FUSED, optional: the prolongation of the coarse-grid correction, xin = x + P xcoarse (interpolate_and_sum_to!,
        // src/interpolation.jl:64-74: identity rows += 1.0 c[a], midpoints += 0.5 c[a], += 0.5 c[b] in the CSC
        // column order), from the cell's coarse column staged in LDS behind the lattice image
        const double *ccol = FUSED && !RS && a.xcoarse ? a.xcoarse + cell * a.ldc : nullptr;
        // (CG, flags bit 6: no room behind the image -- see the in-image path below)
        constexpr bool cgather = FUSED && CG;
        double *cs = xs + nf + lv.lds_g1;
        double cval = 0.0;
        if (ccol && !cgather) {
            if (tid < lv.nf_coarse) cval = ccol[tid];
            for (int q = tid + NT; q < lv.nf_coarse; q += NT) cs[q] = ccol[q];
        }
        auto prolong = [&](double v, uint32_t w) {
            const uint32_t pa = w & 0xffffu, pb = w >> 16;
            if (pa == pb) return v + cs[pa];
            v += 0.5 * cs[pa];
            return v + 0.5 * cs[pb];
        };
        // CG: the coarse column goes to the EVEN nodes of the lattice image (coarse node (i,j,k) = fine node (2i,2j,2k)),
        // every thread combines its slots' parents from there into registers, and only then -- behind a second barrier --
        // the image receives the fine values: no LDS beyond the image (three workgroups per CU stay resident), no gathers
        // from global memory (round 2: two dependent L1/L2 gathers per slot made this the slowest launch of the V-cycle,
        // 4.2 TB/s), the whole column in one batch of loads like the light passes.  Same roundings as the staged form.
        if (cgather && ccol) {
            constexpr int NC = 2;                                   // coarse slots per thread (host: nf_coarse <= NC * NT)
            double cv2[NC];
            int cl2[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int c = tid + q * NT;
                if (c < lv.nf_coarse) {
                    cv2[q] = ccol[c];
                    cl2[q] = lv.clpos[c];
                }
            }
            double xv[SPT];
            uint64_t pw[SPT];
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int t = tid + q * NT;
                if (t < nf) {
                    xv[q] = xc[t];
                    pw[q] = lv.par64[t];
                }
            }
            if (EARLY) finish_tables();
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if (tid + q * NT < lv.nf_coarse) xs[cl2[q]] = cv2[q];
            __syncthreads();                                        // coarse values in place
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int t = tid + q * NT;
                if (t < nf) {
                    const uint32_t pa = (uint32_t)pw[q] & 0xffffu, pb = ((uint32_t)pw[q] >> 16) & 0xffffu;
                    double v = xv[q];
                    if (pa == pb)
                        v = v + lds_ld(xs + pa);
                    else {
                        v += 0.5 * lds_ld(xs + pa);
                        v = v + 0.5 * lds_ld(xs + pb);
                    }
                    xv[q] = v;
                }
            }
            __syncthreads();                                        // every parent read: the image may take the fine values
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int t = tid + q * NT;
                if (t < nf) {
                    const double v = xv[q];
                    if (xoc) xoc[t] = v;
                    rr += v * v;
                    xs[(int)(pw[q] >> 32)] = v;
                }
            }
        }
        // A fused pass that reads nothing but its own column (CG step 0 with r itself as p: no x2 / xacc / x3 / coarse
        // column) takes the whole column in ONE batch like the plain apply -- one memory round trip instead of two.
        // (RS: the local residual with pending CG updates -- never a light pass, never a prolongation)
        const bool light = LF == 1 || (LF == 0 && !RS && FUSED && NT >= 512 && !x2c && !xac && !x3c && !ccol);   // (level 5, 256 threads: the extra path costs a resident workgroup)
        if (FUSED && light) {
            double xv[SPT];
            int lp[SPT];
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int t = tid + q * NT;
                if (t < nf) {
                    xv[q] = xc[t];
                    lp[q] = lv.lpos[t];
                }
            }
            if (EARLY) finish_tables();
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int t = tid + q * NT;
                if (t < nf) {
                    const double v = xv[q];
                    if (xoc) xoc[t] = v;
                    rr += v * v;
                    xs[lp[q]] = v;
                }
            }
        }
        // ... and the dead step of a pre-smoother (reads r and p, forms p_new only in LDS, writes nothing): two streams
        const bool light2 = LF == 2 || (LF == 0 && !RS && FUSED && NT >= 512 && x2c && !xoc && !xac && !x3c && !ccol);
        if (FUSED && light2) {
            double xv[SPT], x2v[SPT];
            int lp[SPT];
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int t = tid + q * NT;
                if (t < nf) {
                    xv[q] = xc[t];
                    x2v[q] = x2c[t];
                    lp[q] = lv.lpos[t];
                }
            }
            if (EARLY && rows_late) issue_rows();
            if (EARLY) finish_tables();
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int t = tid + q * NT;
                if (t < nf) {
                    const double v = axpy1(beta, x2v[q], xv[q]);
                    if (LF == 2 && xoc) xoc[t] = v;      // (round 4: the step that writes its direction next to the old one, lazy_top = 2)
                    rr += v * v;
                    xs[lp[q]] = v;
                }
            }
        }
        constexpr int HB = FUSED ? (SPT + 1) / 2 : SPT;
        // flags bit 7 (fused passes through the batch path below): x is known to be zero -- a coarse level entered with the zero
        // initial guess of src/multigrid.jl:106 -- and is not read (its memory need not even hold the zeros yet)
        // (compiled into the instantiations that can meet it only -- the local residual of a register-blocked level is the RS
        //  one -- : in the main fused instantiation the extra live value cost three spilled registers)
        const bool xzero = FUSED && (RS || RB == 0) && (a.flags & 128);
        if (!light && !light2 && !(cgather && ccol)) {
#pragma unroll
        for (int q0 = 0; q0 < SPT; q0 += HB) {
            double xv[HB], x2v[HB], xav[HB];
            int lp[HB];
            uint32_t pw[HB];
#pragma unroll
            for (int q = 0; q < HB; ++q) {
                const int t = tid + (q0 + q) * NT;
                if (q0 + q < SPT && t < nf) {
                    xv[q] = xzero ? 0.0 : xc[t];
                    x2v[q] = x2c ? x2c[t] : 0.0;
                    xav[q] = xac ? xac[t] : x3c ? x3c[t] : 0.0;
                    lp[q] = lv.lpos[t];
                    if (FUSED && !cgather) pw[q] = ccol ? lv.par32[t] : 0u;
                }
            }
            if (EARLY && q0 == 0) finish_tables();          // (the rows were requested before this batch's loads)
            if (FUSED && q0 == 0 && ccol && !cgather) {   // coarse column complete before the first use
                if (tid < lv.nf_coarse) cs[tid] = cval;
                __syncthreads();
            }
#pragma unroll
            for (int q = 0; q < HB; ++q) {
                const int t = tid + (q0 + q) * NT;
                if (q0 + q < SPT && t < nf) {
                    double v = xv[q];
                    if (FUSED) {
                        if (!cgather && ccol) v = prolong(v, pw[q]);
                        if (xac) xac[t] = axpy1(ax, x2v[q], xav[q]);
                        if (x3c) {
                            const double t1 = axpy1(ax, x2v[q], v);
                            const double p2 = axpy1(beta, x2v[q], xav[q]);
                            v = axpy1(c2, p2, t1);
                        } else if (x2c)
                            v = axpy1(beta, x2v[q], v);
                        if (xoc) xoc[t] = v;
                        if (!RS) rr += v * v;      // (RS: the local residual -- nobody asks for its reductions)
                    }
                    xs[lp[q]] = v;
                }
            }
        }
        }
        for (int t = tid + SPT * NT; t < nf; t += NT) {   // only for cells larger than SPT*NT
            double v = xzero ? 0.0 : xc[t];
            if (FUSED) {
                const double pv = x2c ? x2c[t] : 0.0;
                if (!cgather && ccol) v = prolong(v, lv.par32[t]);
                if (xac) xac[t] = axpy1(ax, pv, xac[t]);
                if (x3c) {
                    const double t1 = axpy1(ax, pv, v);
                    const double p2 = axpy1(beta, pv, x3c[t]);
                    v = axpy1(c2, p2, t1);
                } else if (x2c)
                    v = axpy1(beta, pv, v);
                if (xoc) xoc[t] = v;
                rr += v * v;
            }
            xs[lv.lpos[t]] = v;
        }
    }

    HMG_STAMP(1);   // column in LDS (this wave)
    // (scheduling fence: keeps the table prefetch below out of the load phase's register budget -- the
    //  1024-thread variants must stay within 64 VGPRs to keep two workgroups resident per CU)
    __builtin_amdgcn_sched_barrier(0);
    if (!WEARLY) issue_words();
    if (!EARLY) {
        issue_rows();
        finish_tables();
    }
    const double *sc = a.src ? a.src + cell * lv.ld : nullptr;
    double *oc = a.out ? a.out + cell * lv.ld : nullptr;   // null: only the reductions of a FUSED pass are wanted
    const int nsurf = lv.off_int;
    const int nsw = RB ? lv.nblk : lv.nsweep;
    const bool wdot = WD && FUSED && (a.flags & 8);   // src multiplies: out = alpha A x, pap += mult (x + src) out
    HMG_STAMP(2);   // tables requested, before the barrier
    __syncthreads();
    HMG_STAMP(3);

    // surface entities
    const int wbase = RB ? (1 + lv.nface + lv.nedge) * NDIR : 0;     // RB: the LDS table starts at the corners' rows
    auto surface_node = [&](int t, uint32_t pw) -> double {   // any class: weight row read from the LDS class table tap by tap
        const double sv = sc ? sc[t] : 0.0;
        int L, len, A, B, cls;
        decode32<DIM>(pw, m, L, len, A, B, cls);
        double ctr;
        double o = stencil_eval_c<DIM>(W + cls * NDIR - wbase, xs, L, len, A, B, ctr);
        if (!wdot) o = sv + o;
        if ((dm >> (cls - 1)) & 1u) o = 0.0;
        if (!FUSED || oc) oc[t] = o;
        if (FUSED && !RS) {
            const int e = cls - 1;
            const uint32_t word = e < 4 ? mq[0] : e < 8 ? mq[1] : e < 12 ? mq[2] : mq[3];
            const uint32_t mu = (word >> (8 * (e & 3))) & 0xffu;
            pap += (double)mu * ((wdot ? ctr + sv : ctr) * o);
        }
        return o;
    };
    // RS: every thread keeps what it evaluates (faces FI, edges NE, corner 1, interior NPASS x R values) for the epilogue
    // (dead arrays in every other instantiation)
    double kf[FI], ke[NE], kc = 0.0, ki[NPASS][RB ? RB : 1];
    if constexpr (RS && RB != 0 && DIM == 3) {
        // RS: the interior FIRST -- its sums stay in registers, and the surface runs need fewer registers beside them than the
        // blocked walk would beside the surface results (no reduction of this launch depends on the order)
        double w0e[15];
#pragma unroll
        for (int d = 0; d < NDIR; ++d) w0e[d] = readlane_f64(wv, d);
        if (tid < nsw) interior_block_keep<RB ? RB : 1, FUSED>(w0e, xs, m, nf >> 1, q0, s0, sc, oc, ki[0]);
        if (NPASS > 1 && tid + NT < nsw) interior_block_keep<RB ? RB : 1, FUSED>(w0e, xs, m, nf >> 1, q1, s1, sc, oc, ki[NPASS - 1]);
    }
    if (RB) {
        if (DIM == 3) {
            const bool fdir = (dm >> face) & 1u;
            const double fmult = (double)((mq[0] >> (8 * (face & 3))) & 0xffu);
            const int fbase = lv.off_face + face * lv.nfi;
            double nopre[FI];
#pragma unroll
            for (int q = 0; q < FI; ++q) nopre[q] = 0.0;
            if (face >= 4)
                ;
            else if (face == 0)
                face_items<0, FI, FUSED, FUSED && !RS>(wv, 16, xs, m, lv.nfi, fbase, ft0, fw, fdir, fmult, sc, oc, pap, lane, nopre, false, wdot, RS ? kf : nullptr);
            else if (face == 1)
                face_items<1, FI, FUSED, FUSED && !RS>(wv, 16, xs, m, lv.nfi, fbase, ft0, fw, fdir, fmult, sc, oc, pap, lane, nopre, false, wdot, RS ? kf : nullptr);
            else if (face == 2)
                face_items<2, FI, FUSED, FUSED && !RS>(wv, 16, xs, m, lv.nfi, fbase, ft0, fw, fdir, fmult, sc, oc, pap, lane, nopre, false, wdot, RS ? kf : nullptr);
            else
                face_items<3, FI, FUSED, FUSED && !RS>(wv, 16, xs, m, lv.nfi, fbase, ft0, fw, fdir, fmult, sc, oc, pap, lane, nopre, false, wdot, RS ? kf : nullptr);
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                if (edge[q] < 0) continue;
                const int eb = lv.nface + edge[q];
                const bool edir = (dm >> eb) & 1u;
                const double emult = (double)(((eb < 4 ? mq[0] : eb < 8 ? mq[1] : mq[2]) >> (8 * (eb & 3))) & 0xffu);
                const double nop1[1] = {0.0};
                const int wl = 32 + 16 * q;
                switch (edge[q]) {
                case 0: class_items<edge_tap_mask(0), 1, FUSED, FUSED && !RS>(wv, wl, xs, m, lv.nei, ebase[q], 0, ew[q], edir, emult, sc, oc, pap, lane, nop1, false, wdot, RS ? &ke[q] : nullptr); break;
                case 1: class_items<edge_tap_mask(1), 1, FUSED, FUSED && !RS>(wv, wl, xs, m, lv.nei, ebase[q], 0, ew[q], edir, emult, sc, oc, pap, lane, nop1, false, wdot, RS ? &ke[q] : nullptr); break;
                case 2: class_items<edge_tap_mask(2), 1, FUSED, FUSED && !RS>(wv, wl, xs, m, lv.nei, ebase[q], 0, ew[q], edir, emult, sc, oc, pap, lane, nop1, false, wdot, RS ? &ke[q] : nullptr); break;
                case 3: class_items<edge_tap_mask(3), 1, FUSED, FUSED && !RS>(wv, wl, xs, m, lv.nei, ebase[q], 0, ew[q], edir, emult, sc, oc, pap, lane, nop1, false, wdot, RS ? &ke[q] : nullptr); break;
                case 4: class_items<edge_tap_mask(4), 1, FUSED, FUSED && !RS>(wv, wl, xs, m, lv.nei, ebase[q], 0, ew[q], edir, emult, sc, oc, pap, lane, nop1, false, wdot, RS ? &ke[q] : nullptr); break;
                default: class_items<edge_tap_mask(5), 1, FUSED, FUSED && !RS>(wv, wl, xs, m, lv.nei, ebase[q], 0, ew[q], edir, emult, sc, oc, pap, lane, nop1, false, wdot, RS ? &ke[q] : nullptr); break;
                }
            }
            if (wave == NW - 1 && lane < lv.ncorner) kc = surface_node(lane, cw);
        }
    } else {
        const int nit_surf = (nsurf + NT - 1) / NT;
        for (int it = 0; it < nit_surf; ++it) {
            const int t = tid + it * NT;
            const uint32_t pw = p0;
            p0 = p1;
            p1 = lv.pos32[t + 2 * NT];
            if (t < nsurf) surface_node(t, pw);
        }
    }
    HMG_STAMP(4);
    // cell interior: one weight row for all nodes
    double w0[15];
#pragma unroll
    for (int d = 0; d < NDIR; ++d) w0[d] = RB ? readlane_f64(wv, d) : to_sgpr(W[d]);
    if (RB) {
        // register-blocked interior: one pass (the host selects this instantiation only if nblk <= NT), its word
        // was fetched before the barrier
        if constexpr (!RS) {
        if (DIM == 3 && tid < nsw) {
            if (sc)
                interior_block<RB ? RB : 1, FUSED, true>(w0, xs, m, nf >> 1, q0, s0, sc, oc, pap, wdot);
            else
                interior_block<RB ? RB : 1, FUSED, false>(w0, xs, m, nf >> 1, q0, s0, sc, oc, pap);
        }
        if (DIM == 3 && NPASS > 1 && tid + NT < nsw) {      // second pass (the host selects NT >= nblk / 2)
            if (sc)
                interior_block<RB ? RB : 1, FUSED, true>(w0, xs, m, nf >> 1, q1, s1, sc, oc, pap, wdot);
            else
                interior_block<RB ? RB : 1, FUSED, false>(w0, xs, m, nf >> 1, q1, s1, sc, oc, pap);
        }
        }
    } else {
        const int nit_sweep = (nsw + NT - 1) / NT;
        for (int it = 0; it < nit_sweep; ++it) {
            const int u = tid + it * NT;
            const uint32_t pw = q0;
            const int t = s0;
            q0 = q1;
            s0 = s1;
            q1 = iw[u + 2 * NT];
            s1 = (int)is[u + 2 * NT];
            if (t != 0xffff) {
                const double sv = sc ? sc[t] : 0.0;
                int L, len, A, B, cls;
                decode32<DIM>(pw, m, L, len, A, B, cls);
                double ctr;
                double o = stencil_eval_v<DIM>(w0, xs + L, len, A, B, ctr);
                if (!wdot) o = sv + o;
                if (!FUSED || oc) oc[t] = o;
                if (FUSED) pap += (wdot ? ctr + sv : ctr) * o;
            }
        }
    }
    HMG_STAMP(5);
    if constexpr (RS && RB != 0 && DIM == 3) {
        // Restriction in the epilogue (restrict_to!, src/interpolation.jl:52-62, of the cell-local residual this launch has just
        // formed -- src/multigrid.jl:104-105): the results go from the registers into the lattice image, which nobody reads
        // any more behind the barrier, and the coarse right-hand side is the 15-point sum 1 x own + 0.5 x neighbours at the
        // EVEN lattice nodes, taps that leave the cell weighted 0 -- tap for tap the arithmetic of the stand-alone restriction
        // (k_apply_slab with flags bit 2), so the coarse vector is the same to the last bit.  The residual itself need not go
        // to HBM at all (oc == nullptr): 16 B per fine DOF and one launch less per V-cycle.
        // (the image positions are fetched again here, in front of the barrier that hides their latency, instead of keeping the
        //  addressing words of the evaluation alive across the kernel: held, one of them was spilled, and every reload of a spilled
        //  register waits for ALL outstanding global loads -- vmcnt(0) -- in the middle of the face runs)
        int fl[FI], el[NE], cl = 0;
        uint32_t b0 = 0u, b1 = 0u;
        if (face < 4) {
#pragma unroll
            for (int q = 0; q < FI; ++q) {
                const int ti = ft0 + q * 64 + lane;
                fl[q] = ti < lv.nfi ? (int)lv.lpos[lv.off_face + face * lv.nfi + ti] : 0;
            }
        }
#pragma unroll
        for (int q = 0; q < NE; ++q) el[q] = edge[q] >= 0 && lane < lv.nei ? (int)lv.lpos[ebase[q] + lane] : 0;
        if (wave == NW - 1 && lane < lv.ncorner) cl = (int)lv.lpos[lane];
        if (tid < nsw) b0 = lv.blk_word[tid];
        if (NPASS > 1 && tid + NT < nsw) b1 = lv.blk_word[tid + NT];
        __syncthreads();
        if (face < 4) {
#pragma unroll
            for (int q = 0; q < FI; ++q)
                if (ft0 + q * 64 + lane < lv.nfi) xs[fl[q]] = kf[q];
        }
#pragma unroll
        for (int q = 0; q < NE; ++q)
            if (edge[q] >= 0 && lane < lv.nei) xs[el[q]] = ke[q];
        if (wave == NW - 1 && lane < lv.ncorner) xs[cl] = kc;
        {
            int pos[RB ? RB : 1];
            if (tid < nsw) {
                block_positions<RB ? RB : 1>(m, b0, pos);
                const int nv = (int)(b0 >> 28);
#pragma unroll
                for (int r = 0; r < (RB ? RB : 1); ++r)
                    if (r < nv) xs[pos[r]] = ki[0][r];
            }
            if (NPASS > 1 && tid + NT < nsw) {
                block_positions<RB ? RB : 1>(m, b1, pos);
                const int nv = (int)(b1 >> 28);
#pragma unroll
                for (int r = 0; r < (RB ? RB : 1); ++r)
                    if (r < nv) xs[pos[r]] = ki[NPASS - 1][r];
            }
        }
        __syncthreads();
        double *rc = a.rcoarse + cell * a.ldrc;
        if (lv.rs_lp) {
            // (levels whose stand-alone restriction is k_restrict: its order -- the reference's -- and its roundings)
            for (int c = tid; c < lv.nf_coarse; c += NT) {
                const int b = lv.rptr[c], n = lv.rptr[c + 1] - b;
                int lp[15];
#pragma unroll
                for (int q = 0; q < 15; ++q) lp[q] = lv.rs_lp[b + (q < n ? q : 0)];
                double tmp = 0.0;
                tmp += 1.0 * lds_ld(xs + lp[0]);
#pragma unroll
                for (int q = 1; q < 15; ++q)
                    if (q < n) tmp += 0.5 * lds_ld(xs + lp[q]);
                rc[c] = tmp;
            }
        } else
        for (int c = tid; c < lv.nf_coarse; c += NT) {
            int L, len, A, B, cls, k;
            decode32w(lv.rs_word[c], m, L, len, A, B, cls, k);
            const double *w = lv.rs_w + cls * NDIR;
            auto at = [&](int off) { return lds_ld(xs + max(L + off, 0)); };
            double acc = w[0] * lds_ld(xs + L);
            acc += w[1] * lds_ld(xs + L + 1);
            acc += w[2] * at(-1);
            acc += w[3] * lds_ld(xs + L + len - 1);
            acc += w[4] * at(-len);
            acc += w[5] * lds_ld(xs + L + len);
            acc += w[6] * at(-len - 1);
            acc += w[7] * lds_ld(xs + L + A - len);
            acc += w[8] * at(len + 1 - B);
            acc += w[9] * lds_ld(xs + L + A - 1);
            acc += w[10] * at(1 - B);
            acc += w[11] * lds_ld(xs + L + A);
            acc += w[12] * at(-B);
            acc += w[13] * lds_ld(xs + L + A + 1 - len);
            acc += w[14] * at(len - B);
            rc[c] = 0.0 + acc;               // (the stand-alone kernel adds its absent source value first: 0 + acc)
        }
    }
    if (FUSED && !RS) {
        __syncthreads();                     // W / xs no longer read: reuse the front of LDS for the reduction
        const double s_pap = block_sum(pap, smem);
        const double s_rr = block_sum(rr, smem);
        if (tid == 0) {
            a.blockpart[2 * cell] = s_pap;
            a.blockpart[2 * cell + 1] = s_rr;
        }
    }
    HMG_STAMP(6);
    if (LOOP) {
        blk += gridDim.x;
        if (blk >= a.nwork) break;
        __syncthreads();                         // the next cell's table and image overwrite this one's
    }
    } while (LOOP);
}

// Slab variant for cells whose lattice image exceeds the LDS (level 7 in 3D: 374 KiB).  The cell is processed in
// slabs of consecutive k-planes through a ROLLING window: the LDS holds planes [k0-1, k1] of the lattice image, the
// nodes of planes [k0, k1) are evaluated, then planes k1-1 and k1 are moved to the front of the window (LDS -> LDS)
// and only planes k1+1 .. k1' come from HBM -- every slot is read from HBM exactly once, so the load-phase side
// effects of the fused CG pass (p-update, x-update, r.r) can be attached to it as in k_apply.  Per slab the host
// provides two flat lists (SlabTables): the slots that are new in the window (LDS position | slot << 16) and the
// slots that are evaluated (addressing word + slot; surface entities first, then the cell interior), so that both
// phases are single batched / software-pipelined loops instead of one latency-bound loop per entity.  The window
// is sized so that two workgroups are resident per CU (one loads while the other computes).  Same arithmetic as
// k_apply.
template <int DIM, int NT, bool FUSED, bool WD = false>
__global__ void __launch_bounds__(NT, 8)
k_apply_slab(LevelDev lv, const double *__restrict__ coef, const uint16_t *__restrict__ dmask, ApplyArgs a, SlabTables st)
{
    constexpr int NDIR = DIM == 3 ? 15 : 7;
    constexpr int NTERM = DIM == 3 ? 7 : 4;
    constexpr int HB = FUSED ? 3 : 4;               // loads in flight per thread and stream in the load phase (64-VGPR budget)
    constexpr int MV = 5;                           // values per thread of the window move (two planes of level 7: <= 4225 nodes)
    extern __shared__ double smem[];
    double *W = smem;
    double *img = smem + WSZ;                       // lds_nodes doubles: [planes k0-1..k1 | zero guard]
    const int tid = threadIdx.x;
    const int64_t cell = a.cell_list ? (int64_t)a.cell_list[blockIdx.x] : (int64_t)blockIdx.x;
    const int m = lv.m;

    if (a.flags & 4) {   // cell-independent stencil (restriction): the weight is the last term of the class table
        for (int idx = tid; idx < lv.ncls * NDIR; idx += NT) W[idx] = lv.ctab[(size_t)idx * NTERM + NTERM - 1];
    } else {
        double s[NTERM];
        cell_scales<DIM>(coef + cell * 8, a.alpha, a.lambda, s, a.flags);
        for (int idx = tid; idx < lv.ncls * NDIR; idx += NT) {
            const double *c = lv.ctab + (size_t)idx * NTERM;
            double w = 0.0;
#pragma unroll
            for (int t = 0; t < NTERM; ++t) w += c[t] * s[t];
            W[idx] = w;
        }
    }
    const double *xc = a.x + cell * lv.ld;
    const double *x2c = FUSED && a.x2 ? a.x2 + cell * lv.ld : nullptr;
    double *xoc = FUSED && a.xout ? a.xout + cell * lv.ld : nullptr;
    double *xac = FUSED && a.xacc ? a.xacc + cell * lv.ld : nullptr;
    const double *x3c = FUSED && a.x3 ? a.x3 + cell * lv.ld : nullptr;   // two pending x-updates, see k_apply
    const double beta = to_sgpr(x2c ? a.scal[a.s_num] / a.scal[a.s_den] : 0.0);
    const double ax = to_sgpr(xac || x3c ? a.scal[a.a_num] / a.scal[a.a_den] : 0.0);
    const double c2 = to_sgpr(x3c ? a.scal[a.c_num] / a.scal[a.c_den] : 0.0);
    const uint32_t dm = (a.flags & 1) ? dmask[cell] : 0u;
    const double *sc = a.src ? a.src + cell * lv.ld : nullptr;
    double *oc = a.out ? a.out + cell * (a.out_ld ? a.out_ld : (int64_t)lv.ld) : nullptr;
    uint32_t mq[4] = {0, 0, 0, 0};   // the cell's 16 entity multiplicities, wave-uniform -> SGPRs
    if (FUSED) {
        const uint32_t *mp = reinterpret_cast<const uint32_t *>(a.mult + cell * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) mq[q] = a.mult ? __builtin_amdgcn_readfirstlane(mp[q]) : 0x01010101u;
    }
    const bool wdot = WD && FUSED && (a.flags & 8);   // src multiplies: out = alpha A x, pap += mult (x + src) out
    double rr = 0.0, pap = 0.0;
    auto plane_off = [&](int k) {   // PO(k) = number of lattice nodes in planes < k
        if (k <= 0) return 0;
        if (k > m + 1) k = m + 1;
        const long long n1 = m + 1, n2 = m + 1 - k;
        return (int)((n1 * (n1 + 1) * (n1 + 2) - n2 * (n2 + 1) * (n2 + 2)) / 6);
    };
    HMG_SLAB_T0();
    __syncthreads();                                                // W complete
    HMG_SLAB_PHASE(0);                                              // (0: weight table)
    double w0[NDIR];                                                // interior weight row, SGPR-resident
#pragma unroll
    for (int d = 0; d < NDIR; ++d) w0[d] = to_sgpr(W[d]);
    int lo_prev = 0;
    for (int sl = 0; sl < st.nslab; ++sl) {
        const int *hd = st.head + 8 * sl;                           // k0, ld_off, ld_cnt, cp_off, cp_cnt, cp_surf
        const int k0 = hd[0], ld_off = hd[1], ld_cnt = hd[2], cp_off = hd[3], cp_cnt = hd[4], cp_surf = hd[5];
        const int lo = plane_off(k0 - 1);                           // lattice range held in LDS: [lo, PO(k1 + 1))
        double *xs = img - lo;                                      // xs[L] valid inside that range (+ zero guard)
        // (round 4) the words of the slab's first batch of loads are requested before the window is rearranged: their L2 round
        // trip -- the column loads depend on them -- runs behind the move and the zero fill instead of after them
        uint32_t wd0[HB];
#pragma unroll
        for (int q = 0; q < HB; ++q) wd0[q] = st.ld_word[ld_off + q * NT + tid];
        if (sl > 0) {
            __syncthreads();                                        // previous slab fully consumed
            // planes k0-1 and k0 (the last evaluated plane and the upper halo of the previous slab) move to the
            // front of the window; destination below source, ascending chunks, one barrier between a chunk's
            // reads and its writes
            const int cnt = plane_off(k0 + 1) - lo, src = lo - lo_prev;
            if (cnt <= MV * NT) {
                // (round 4) every thread takes its share of the two planes into registers, ONE barrier, then writes: the chunked form
                // below pays a barrier per 1024 values -- five per slab near the base of the cell
                double mv[MV];
#pragma unroll
                for (int c = 0; c < MV; ++c) {
                    const int q = c * NT + tid;
                    mv[c] = q < cnt ? img[src + q] : 0.0;
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < MV; ++c) {
                    const int q = c * NT + tid;
                    if (q < cnt) img[q] = mv[c];
                }
            } else {
            const int nchunk = (cnt + NT - 1) / NT;
            for (int c = 0; c < nchunk; ++c) {
                const int q = c * NT + tid;
                const double v = q < cnt ? img[src + q] : 0.0;
                __syncthreads();
                if (q < cnt) img[q] = v;
            }
            }
        }
        for (int q = (sl > 0 ? plane_off(k0 + 1) - lo : 0) + tid; q < st.lds_nodes; q += NT) img[q] = 0.0;   // stale data + guard
        __syncthreads();
        HMG_SLAB_PHASE(1);                                          // (1: window move + zero fill, with the wait for the previous slab's consumers)
        // planes new in the window: HBM -> LDS, every slot once; batches of HB slots per thread, all loads of a
        // batch before its stores (xout / xacc may alias x2)
        for (int q0 = 0; q0 < ld_cnt; q0 += HB * NT) {
            uint32_t wd[HB];
            double xv[HB], x2v[HB], xav[HB];
#pragma unroll
            for (int q = 0; q < HB; ++q) {
                const int v = q0 + q * NT + tid;
                wd[q] = q0 == 0 ? wd0[q] : st.ld_word[ld_off + v];   // (list is padded: no bounds check)
            }
#pragma unroll
            for (int q = 0; q < HB; ++q) {
                const int v = q0 + q * NT + tid;
                if (v < ld_cnt) {
                    const int t = (int)(wd[q] >> 16);
                    xv[q] = xc[t];
                    x2v[q] = x2c ? x2c[t] : 0.0;
                    xav[q] = xac ? xac[t] : x3c ? x3c[t] : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < HB; ++q) {
                const int v = q0 + q * NT + tid;
                if (v < ld_cnt) {
                    const int t = (int)(wd[q] >> 16);
                    double val = xv[q];
                    if (FUSED) {
                        if (xac) xac[t] = axpy1(ax, x2v[q], xav[q]);
                        if (x3c) {
                            const double t1 = axpy1(ax, x2v[q], val);
                            const double p2 = axpy1(beta, x2v[q], xav[q]);
                            val = axpy1(c2, p2, t1);
                        } else if (x2c)
                            val = axpy1(beta, x2v[q], val);
                        if (xoc) xoc[t] = val;
                        rr += val * val;
                    }
                    xs[wd[q] & 0xffffu] = val;
                }
            }
        }
        // evaluated planes, surface entities first: addressing word and slot fetched two iterations ahead of use
        const int ib = cp_off + cp_surf;                            // start of the interior part of the list
        uint32_t p0 = st.cp_word[cp_off + tid], p1 = st.cp_word[cp_off + tid + NT];
        int s0 = (int)st.cp_slot[cp_off + tid], s1 = (int)st.cp_slot[cp_off + tid + NT];
        uint32_t q0 = st.cp_word[ib + tid], q1 = st.cp_word[ib + tid + NT];
        int t0 = (int)st.cp_slot[ib + tid], t1 = (int)st.cp_slot[ib + tid + NT];
        __syncthreads();
        HMG_SLAB_PHASE(2);                                          // (2: HBM -> LDS, to the barrier behind it)
        const int nit_s = (cp_surf + NT - 1) / NT;
        for (int it = 0; it < nit_s; ++it) {
            const int v = it * NT + tid;
            const uint32_t pw = p0;
            const int t = s0;
            p0 = p1;
            s0 = s1;
            p1 = st.cp_word[cp_off + v + 2 * NT];
            s1 = (int)st.cp_slot[cp_off + v + 2 * NT];
            if (v < cp_surf) {
                const double sv = sc ? sc[t] : 0.0;
                int L, len, A, B, cls, k;
                decode32w(pw, m, L, len, A, B, cls, k);
                double ctr;
                double o = stencil_eval_c<DIM>(W + cls * NDIR, xs, L, len, A, B, ctr);
                if (!wdot) o = sv + o;
                if ((dm >> (cls - 1)) & 1u) o = 0.0;
                if (!FUSED || oc) oc[t] = o;
                if (FUSED) {
                    const int en = cls - 1;
                    const uint32_t word = en < 4 ? mq[0] : en < 8 ? mq[1] : en < 12 ? mq[2] : mq[3];
                    const uint32_t mu = (word >> (8 * (en & 3))) & 0xffu;
                    pap += (double)mu * ((wdot ? ctr + sv : ctr) * o);
                }
            }
        }
        HMG_SLAB_PHASE(3);                                          // (3: surface entities)
        const int n_int = cp_cnt - cp_surf;
        const int nit_i = (n_int + NT - 1) / NT;
        for (int it = 0; it < nit_i; ++it) {                        // cell interior: one weight row for all nodes
            const int v = it * NT + tid;
            const uint32_t pw = q0;
            const int t = t0;
            q0 = q1;
            t0 = t1;
            q1 = st.cp_word[ib + v + 2 * NT];
            t1 = (int)st.cp_slot[ib + v + 2 * NT];
            if (v < n_int) {
                const double sv = sc ? sc[t] : 0.0;
                int L, len, A, B;
                decode_lattice(pw, m, L, len, A, B);
                double ctr;
                double o = stencil_eval_v<DIM>(w0, xs + L, len, A, B, ctr);
                if (!wdot) o = sv + o;
                if (!FUSED || oc) oc[t] = o;
                if (FUSED) pap += (wdot ? ctr + sv : ctr) * o;
            }
        }
        lo_prev = lo;
        HMG_SLAB_PHASE(4);                                          // (4: cell interior)
    }
    HMG_SLAB_STORE();
    if (FUSED) {
        __syncthreads();
        const double s_pap = block_sum(pap, smem);
        const double s_rr = block_sum(rr, smem);
        if (tid == 0) {
            a.blockpart[2 * cell] = s_pap;
            a.blockpart[2 * cell + 1] = s_rr;
        }
    }
}

// blockpart[2*c + {0,1}] -> partials[b], partials[2048 + b]  (256 blocks, fixed order: deterministic)
__global__ void __launch_bounds__(256)
k_reduce_pairs(const double *__restrict__ blockpart, int64_t n, double *partials)
{
    __shared__ double red[4];
    double s0 = 0.0, s1 = 0.0;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n; c += (int64_t)gridDim.x * 256) {
        s0 += blockpart[2 * c];
        s1 += blockpart[2 * c + 1];
    }
    const double r0 = block_sum(s0, red);
    const double r1 = block_sum(s1, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = r0;
        partials[2048 + blockIdx.x] = r1;
    }
}

// driver integrals: sum over the first n cells of |J_c| * blockpart[2 c]  (|J| = last term of the cell's coefficients)
__global__ void __launch_bounds__(256)
k_reduce_weighted(const double *__restrict__ blockpart, const double *__restrict__ coef, int jterm, int64_t n, double *partials)
{
    __shared__ double red[4];
    double s0 = 0.0;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n; c += (int64_t)gridDim.x * 256)
        s0 += blockpart[2 * c] * coef[c * 8 + jterm];
    const double r0 = block_sum(s0, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = r0;
}

size_t apply_lds_bytes(const LevelDev &lv)
{
    return sizeof(double) * (size_t)(WSZ + lv.lds_g0 + lv.nf + lv.lds_g1);
}

bool apply_restricts(const Launch &L, const LevelDev &lv)
{
    // the conditions under which launch_apply_dim reaches the 512-thread register-blocked instantiations (level 6)
    if (!(lv.dim == 3 && apply_lds_bytes(lv) <= 160 * 1024 && L.apply_threads == 0 && !L.apply_unblocked && lv.rs_word &&
          lv.rs_w && lv.nf_coarse > 0 && lv.nei <= 64 && lv.nedge == 6 && lv.ncorner == 4))
        return false;
    if (lv.nf > 2048)       // level 6: the 512-thread shape
        return lv.blk_R == 6 && lv.nblk <= 960 && L.apply_wg512 && lv.nfi <= 512;
    // level 5: the 256-thread shape (its stand-alone restriction is k_restrict: the epilogue needs that kernel's lists)
    return lv.nf > 192 && lv.nf <= 1024 && lv.blk_R == 4 && lv.nblk <= 192 && lv.nfi <= 128 && lv.rs_lp != nullptr;
}

static size_t apply_lds_bytes_rb(const LevelDev &lv)   // register-blocked instantiations: only the corners' weight rows in LDS
{
    return sizeof(double) * (size_t)(WSZ_RB + lv.lds_g0 + lv.nf + lv.lds_g1);
}

// Every device base a kernel dereferences without a test of its own must be there: the kernels guard the OPTIONAL operands
// (out of a fused pass, src, x2, xout, xacc, x3, xcoarse), not these.  (Round 2 lost a GPU box to a store through a null
// base 0x4000 bytes in -- an experimental launch path without this check; a host throw costs nothing.)
template <bool FUSED>
static void check_apply_bases(const ApplyArgs &a, const MeshDev &mesh)
{
    if (!a.x) throw std::runtime_error("operator apply: null input vector");
    if (!FUSED && !a.out) throw std::runtime_error("operator apply: a plain launch needs an output vector");
    if (a.rcoarse && a.ldrc <= 0) throw std::runtime_error("operator apply: epilogue restriction without the coarse column stride");
    if (FUSED && (!a.blockpart || !a.scal)) throw std::runtime_error("operator apply: fused launch without its reduction scratch");
    if (!mesh.coef) throw std::runtime_error("operator apply: no operator coefficients on the device (hmg_grid_set_operator)");
    if ((a.flags & 1) && !mesh.dmask) throw std::runtime_error("operator apply: constraint requested without a Dirichlet mask");
    if (a.xcoarse && !a.xout) throw std::runtime_error("operator apply: folded prolongation without xout");
    if ((a.flags & 128) && !(FUSED && a.x3 && a.x2 && a.xout && !a.xcoarse))
        throw std::runtime_error("operator apply: the zero-input form exists for the residual with two pending x-updates only");
}

// can this launch take its class weights from the class-weight cache?  (formed for |alpha| = 1, the grid's lambda, the full operator)
static bool weight_cache_ok(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a)
{
    return L.weight_cache && lv.wcache && mesh.cell_class && lv.dim == 3 && lv.ncls == 15 && !(a.flags & (2 | 8 | 16)) &&
           (a.alpha == 1.0 || a.alpha == -1.0) && a.lambda == mesh.wc_lambda;
}

template <int DIM, int NT, int SPT, bool FUSED, int RB = 0, bool WD = false, bool CG = false, bool RS = false, bool WC = false, int LF = 0>
static void launch_apply_generic(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a, size_t lds)
{
    auto kern = k_apply<DIM, NT, SPT, FUSED, RB, WD, CG, RS, WC, LF>;
    if (a.rcoarse && !RS) throw std::runtime_error("operator apply: this instantiation cannot restrict in its epilogue");
    if ((a.flags & 128) && !(FUSED && (RS || RB == 0)))
        throw std::runtime_error("operator apply: this instantiation cannot take a zero input that is not in memory");
    if (FUSED && a.xcoarse && !CG) lds += sizeof(double) * (size_t)lv.nf_coarse;   // coarse column behind the lattice image
    if (lds > 48 * 1024)
        HMG_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t nblocks = a.cell_list ? a.ncell_list : a.ncells_prefix ? a.ncells_prefix : mesh.ncells;
    if (nblocks == 0) return;
    check_apply_bases<FUSED>(a, mesh);
    ApplyArgs b = a;
    b.nwork = nblocks;
    b.cell_class = WC ? mesh.cell_class : nullptr;
    if (WC && !weight_cache_ok(L, lv, mesh, a)) throw std::runtime_error("operator apply: class-weight cache not usable for this launch");
    if (L.cell_order && RB && !a.cell_list && !a.ncells_prefix && mesh.cell_perm) {
        b.cell_list = mesh.cell_perm;              // (XCD x walks the x-th eighth of the cells, see upload_mesh)
        b.ncell_list = nblocks;
    }
    // (NT == 64: persistent one-wave workgroups, see the kernel -- 32 waves per CU are resident)
    const int64_t grid = NT == 64 ? std::min<int64_t>(nblocks, (int64_t)L.persistent_waves) : nblocks;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, L.stream, lv, mesh.coef, mesh.dmask, b);
    check_launch();
}

template <int DIM, bool FUSED, bool WD = false>
static void launch_apply_dim(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a)
{
    const size_t lds = apply_lds_bytes(lv);
    if constexpr (DIM == 3 && !WD) {
        // level 5: one wave per cell, class weights from the cache (hmg_apply_wave.hip)
        if (apply_wave_ok(L, lv, mesh, a, FUSED)) {
            launch_apply_wave(L, lv, mesh, a, FUSED);
            return;
        }
        // levels 2-4: one persistent, software-pipelined wave per cell (hmg_apply_small.hip)
        if (apply_small_ok(L, lv, mesh, a, FUSED)) {
            launch_apply_small(L, lv, mesh, a, FUSED);
            return;
        }
    }
    // (experiment of round 5, option slab2_force + HMG_SLAB_LDS_KB <= 30: level 6 through the role-split window kernel of level 7 --
    //  VERDICT r4 item 4's "cell as two half-images, the second half's loads in flight while the first is evaluated")
    const bool force_slab2 = DIM == 3 && !WD && L.slab2_force && mesh.slab.head && mesh.slab.nslab >= 2 && lv.nf > 2048 &&
                             !a.xcoarse && !a.rcoarse && !(a.flags & (4 | 8));
    if (lds > 160 * 1024 || force_slab2) {
        if (DIM != 3 || !mesh.slab.head) throw std::runtime_error("operator apply: cell does not fit the LDS");
        if constexpr (DIM == 3 && !WD) {
            if (apply_slab2_ok(L, lv, mesh, a)) {
                launch_apply_slab2(L, lv, mesh, a, FUSED);
                return;
            }
        }
        auto kern = k_apply_slab<3, 1024, FUSED, WD>;
        const size_t bytes = sizeof(double) * (size_t)(WSZ + mesh.slab.lds_nodes);
        HMG_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        const int64_t nblocks = a.cell_list ? a.ncell_list : a.ncells_prefix ? a.ncells_prefix : mesh.ncells;
        if (nblocks == 0) return;
        check_apply_bases<FUSED>(a, mesh);
        ApplyArgs b = a;
        b.nwork = nblocks;
        if (L.cell_order && !a.cell_list && !a.ncells_prefix && mesh.cell_perm) {   // (XCD-aware cell order, as in launch_apply_generic)
            b.cell_list = mesh.cell_perm;
            b.ncell_list = nblocks;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(1024), bytes, L.stream, lv, mesh.coef, mesh.dmask, b,
                           mesh.slab);
        check_launch();
        return;
    }
    const int nf = lv.nf;
    int nt = WD ? 0 : L.apply_threads;   // (the integral instantiations exist for the automatic workgroup sizes only)
    // (small cells are bound by the number of waves launched, not by their work: as few waves per cell as hold it)
    // (level 4, 165 nodes: one wave per cell in three passes beats three waves -- 32 instead of 10 cells in flight per CU,
    //  V-cycle from level 4 down 11.26 -> 10.05 ms)
    if (nt == 0) nt = nf <= 192 ? 64 : nf <= 2048 ? 256 : 1024;
    if (nt <= 64) {
        // (level 4, 165 nodes per wave: three slots per lane loaded as batches, not one dependent round trip per slot)
        if constexpr (DIM == 3 && !WD) {
            if (weight_cache_ok(L, lv, mesh, a)) {           // the cell's class table from the class-weight cache
                if (nf > 64)
                    launch_apply_generic<DIM, 64, 3, FUSED, 0, false, false, false, DIM == 3>(L, lv, mesh, a, lds);
                else
                    launch_apply_generic<DIM, 64, 1, FUSED, 0, false, false, false, DIM == 3>(L, lv, mesh, a, lds);
                return;
            }
        }
        if (nf > 64)
            launch_apply_generic<DIM, 64, 3, FUSED, 0, WD>(L, lv, mesh, a, lds);
        else
            launch_apply_generic<DIM, 64, 1, FUSED, 0, WD>(L, lv, mesh, a, lds);
    }
    else if (nt <= 192 && nf <= 192)
        launch_apply_generic<DIM, 192, 1, FUSED, 0, WD>(L, lv, mesh, a, lds);
    else if (nt <= 256 && nf <= 1024) {
        if (DIM == 3 && lv.blk_R == 4 && lv.nblk <= 192 && lv.nfi <= 128 && lv.nei <= 64 && lv.nedge == 6 && lv.ncorner == 4 &&
            !L.apply_unblocked) {
            if constexpr (!WD) {
                if (FUSED && DIM == 3 && a.rcoarse && lv.rs_word && lv.rs_w && !a.xcoarse) {
                    launch_apply_generic<DIM, 256, 4, FUSED, DIM == 3 ? 4 : 0, false, false, FUSED && DIM == 3>(
                        L, lv, mesh, a, apply_lds_bytes_rb(lv));
                    return;
                }
            }
            launch_apply_generic<DIM, 256, 4, FUSED, DIM == 3 ? 4 : 0, WD>(L, lv, mesh, a, apply_lds_bytes_rb(lv));
        }
        else
            launch_apply_generic<DIM, 256, 4, FUSED, 0, WD>(L, lv, mesh, a, lds);
    }
    else if (nt <= 256)
        launch_apply_generic<DIM, 256, 8, FUSED, 0, WD>(L, lv, mesh, a, lds);
    else if (nt <= 640) {
        if constexpr (!WD) {
            if (nt <= 512) {
                if (DIM == 3 && lv.blk_R == 6 && lv.nblk <= 1024 && lv.nfi <= 512 && lv.nei <= 64 && lv.nedge == 6 &&
                    lv.ncorner == 4 && !L.apply_unblocked)
                    launch_apply_generic<DIM, 512, 13, FUSED, DIM == 3 ? 6 : 0>(L, lv, mesh, a, apply_lds_bytes_rb(lv));
                else
                    launch_apply_generic<DIM, 512, 13, FUSED>(L, lv, mesh, a, lds);
            } else
                launch_apply_generic<DIM, 640, 11, FUSED>(L, lv, mesh, a, lds);
        }
    } else if (DIM == 3 && lv.blk_R == 6 && lv.nblk <= 960 && lv.nei <= 64 && lv.nedge == 6 && lv.ncorner == 4 && !L.apply_unblocked) {
        if constexpr (!WD) {
            if (L.apply_wg512 && lv.nfi <= 512) {
                // (own instantiation: the in-image staging of the coarse column costs the others registers; it handles the
                //  residual of the coarse-grid correction only -- no pending CG updates -- and one batch of 13 slots)
                const bool cg = FUSED && DIM == 3 && a.xcoarse && (a.flags & 64) && lv.par64 && lv.clpos && !a.x2 && !a.xacc && !a.x3 &&
                                lv.nf <= 13 * 512 && lv.nf_coarse <= 2 * 512;
                const bool rs = !cg && FUSED && DIM == 3 && a.rcoarse && lv.rs_word && lv.rs_w && !a.xcoarse;   // (own instantiation again)
                constexpr int R6 = DIM == 3 ? 6 : 0;
                constexpr bool F3 = FUSED && DIM == 3;
                if (DIM == 3 && weight_cache_ok(L, lv, mesh, a)) {      // class weights from the cache (WC), else combined per cell
                    if (cg)
                        launch_apply_generic<DIM, 512, 13, FUSED, R6, false, F3, false, DIM == 3>(L, lv, mesh, a, apply_lds_bytes_rb(lv));
                    else if (rs)
                        launch_apply_generic<DIM, 512, 13, FUSED, R6, false, false, F3, DIM == 3>(L, lv, mesh, a, apply_lds_bytes_rb(lv));
                    else if (FUSED && !a.x2 && !a.xacc && !a.x3 && !a.xcoarse && !a.rcoarse)                 // CG step 0
                        launch_apply_generic<DIM, 512, 13, FUSED, R6, false, false, false, DIM == 3, F3 ? 1 : 0>(L, lv, mesh, a, apply_lds_bytes_rb(lv));
                    else if (FUSED && a.x2 && (!a.xout || a.xout != a.x2) && !a.xacc && !a.x3 && !a.xcoarse && !a.rcoarse)   // a dead step, or one that writes its direction elsewhere (no store may sit between the loads of a stream it aliases)
                        launch_apply_generic<DIM, 512, 13, FUSED, R6, false, false, false, DIM == 3, F3 ? 2 : 0>(L, lv, mesh, a, apply_lds_bytes_rb(lv));
                    else
                        launch_apply_generic<DIM, 512, 13, FUSED, R6, false, false, false, DIM == 3>(L, lv, mesh, a, apply_lds_bytes_rb(lv));
                    return;
                }
                if (cg)
                    launch_apply_generic<DIM, 512, 13, FUSED, R6, false, F3>(L, lv, mesh, a, apply_lds_bytes_rb(lv));
                else if (rs)
                    launch_apply_generic<DIM, 512, 13, FUSED, R6, false, false, F3>(L, lv, mesh, a, apply_lds_bytes_rb(lv));
                else
                    launch_apply_generic<DIM, 512, 13, FUSED, R6>(L, lv, mesh, a, apply_lds_bytes_rb(lv));
                return;
            }
        }
        launch_apply_generic<DIM, 1024, 7, FUSED, DIM == 3 ? 6 : 0, WD>(L, lv, mesh, a, apply_lds_bytes_rb(lv));
    }
    else
        launch_apply_generic<DIM, 1024, 7, FUSED, 0, WD>(L, lv, mesh, a, lds);
}

void launch_apply(const Launch &L, const LevelDev &lv, const MeshDev &mesh, double alpha, double lambda,
                  const double *x, const double *src, double *out, int use_mask)
{
    ApplyArgs a{};
    a.alpha = alpha;
    a.lambda = lambda;
    a.x = x;
    a.src = src;
    a.out = out;
    a.flags = (use_mask ? 1 : 0) | (L.apply_mass_only ? 2 : 0);
#ifdef HMG_PHASE_TIMING
    a.blockpart = mesh.blockpart;
#endif
    if (lv.dim == 3)
        launch_apply_dim<3, false>(L, lv, mesh, a);
    else
        launch_apply_dim<2, false>(L, lv, mesh, a);
}

void launch_apply_args(const Launch &L, const LevelDev &lv, const MeshDev &mesh, ApplyArgs a)
{
    a.flags = (a.flags & 1) | (L.apply_mass_only ? 2 : 0);
    if (lv.dim == 3)
        launch_apply_dim<3, false>(L, lv, mesh, a);
    else
        launch_apply_dim<2, false>(L, lv, mesh, a);
}

void launch_apply_fused_kernel(const Launch &L, const LevelDev &lv, const MeshDev &mesh, ApplyArgs a)
{
    a.scal = L.scal;
    if (!(a.flags & 32)) a.mult = mesh.mult;     // (bit 5: caller wants unit multiplicities -- a.mult stays null)
    a.blockpart = mesh.blockpart;
    if (a.flags & 8) {                            // the driver integrals: src multiplies (own instantiations)
        if (lv.dim == 3)
            launch_apply_dim<3, true, true>(L, lv, mesh, a);
        else
            launch_apply_dim<2, true, true>(L, lv, mesh, a);
        return;
    }
    if (lv.dim == 3)
        launch_apply_dim<3, true>(L, lv, mesh, a);
    else
        launch_apply_dim<2, true>(L, lv, mesh, a);
}

void launch_apply_fused_reduce(const Launch &L, const MeshDev &mesh, int slot_pap, int slot_rr)
{
    hipLaunchKernelGGL(k_reduce_pairs, dim3(256), dim3(256), 0, L.stream, mesh.blockpart, mesh.ncells, L.partials);
    check_launch();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, L.stream, L.partials, 256, L.scal, slot_pap);
    if (slot_rr >= 0)
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, L.stream, L.partials + 2048, 256, L.scal, slot_rr);
    check_launch();
}

// ---------------------------------------------------------------------------------------------
// interface sum: every shared entity is one contiguous, identically ordered run in each copy
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_iface_faces(const int32_t *__restrict__ pairs, int64_t npairs, int nfi, int off_face, int ld, double *x)
{
    // one wave per shared face; all loads of (up to) 8 x 64 DOFs of both copies in flight before the first store
    constexpr int U = 8;
    const int lane = threadIdx.x & 63;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; p < npairs; p += nw) {
        const int32_t ca = pairs[3 * p], cb = pairs[3 * p + 1], lf = pairs[3 * p + 2];
        double *a = x + (int64_t)ca * ld + off_face + (lf & 15) * nfi;
        double *b = x + (int64_t)cb * ld + off_face + (lf >> 4) * nfi;
        for (int k0 = 0; k0 < nfi; k0 += U * 64) {
            double va[U], vb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * 64 + lane;
                if (k < nfi) {
                    va[u] = a[k];
                    vb[u] = b[k];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * 64 + lane;
                if (k < nfi) {
                    const double s = va[u] + vb[u];   // (0 + a) + b, ascending cell order
                    a[k] = s;
                    b[k] = s;
                }
            }
        }
    }
}

// CSR entities in one launch: shared edges (nei values at offset off_edge + lid*nei) followed by shared nodes (one
// value at offset lid).  Copies are summed in list order (ascending cell, the reference's order).
__global__ void __launch_bounds__(256)
k_iface_csr(const int32_t *__restrict__ eptr, const int32_t *__restrict__ eent, int64_t nedge, int nei, int off_edge,
            const int32_t *__restrict__ nptr, const int32_t *__restrict__ nent, int64_t nnode, int ld, double *x)
{
    // Copies are read in batches of U: all entries of a batch first, then all values (independent loads in flight together
    // instead of one dependent entry -> value round trip per copy: a shared edge of the Kuhn lattice has 4 or 6 copies, a
    // node 24), summed in list order as before.
    constexpr int U = 8;
    const int64_t tedge = nedge * nei, total = tedge + nnode;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const bool is_edge = idx < tedge;
        const int64_t e = is_edge ? idx / nei : idx - tedge;
        const int k = is_edge ? (int)(idx - e * nei) : 0;
        const int per = is_edge ? nei : 1, off = is_edge ? off_edge : 0;
        const int32_t *ptr = is_edge ? eptr : nptr, *ent = is_edge ? eent : nent;
        const int b = ptr[e], end = ptr[e + 1];
        double s = 0.0;
        if (end - b <= U) {
            // one batch (a shared edge has 4-6 copies): the entries stay in registers for the stores -- the general path below reads
            // them a second time, one more dependent round trip per entity (round 4)
            int32_t v[U];
            double xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = b + u < end ? ent[b + u] : -1;
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = v[u] >= 0 ? x[(int64_t)(v[u] >> 3) * ld + off + (v[u] & 7) * per + k] : 0.0;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (v[u] >= 0) s += xv[u];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (v[u] >= 0) x[(int64_t)(v[u] >> 3) * ld + off + (v[u] & 7) * per + k] = s;
            continue;
        }
        for (int q0 = b; q0 < end; q0 += U) {
            int32_t v[U];
            double xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = q0 + u < end ? ent[q0 + u] : -1;
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = v[u] >= 0 ? x[(int64_t)(v[u] >> 3) * ld + off + (v[u] & 7) * per + k] : 0.0;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (v[u] >= 0) s += xv[u];
        }
        for (int q0 = b; q0 < end; q0 += U) {
            int32_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = q0 + u < end ? ent[q0 + u] : -1;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (v[u] >= 0) x[(int64_t)(v[u] >> 3) * ld + off + (v[u] & 7) * per + k] = s;
        }
    }
}

void launch_interface_sum(const Launch &L, const LevelDev &lv, const MeshDev &mesh, double *x, int which, bool faces)
{
    const int cap = L.num_cu * 8;
    if (lv.dim == 3 && lv.nfi > 0 && mesh.nfacepairs > 0) {
        // pairs [0, ncut) count as cut (rehearsal partitions): summed with the cut groups, whatever `faces` says
        const int64_t ncut = mesh.ncut_face_pairs;
        const int64_t first = which == 2 ? ncut : 0;
        const int64_t last = which == 1 || !faces ? ncut : mesh.nfacepairs;
        const int64_t np = last - first;
        if (np > 0) {
            const int64_t blocks = (np + 3) / 4;              // one wave per pair, blocks dispatched in pair order
            hipLaunchKernelGGL(k_iface_faces, dim3((unsigned)blocks), dim3(256), 0, L.stream, mesh.face_pairs + 3 * first, np,
                               lv.nfi, lv.off_face, lv.ld, x);
            check_launch();
        }
    }
    // group ranges of the CSR lists: [0, ncut) are cut by the partition, [ncut, n) are not
    auto range = [&](int64_t n, int64_t ncut, int64_t &first, int64_t &count) {
        first = which == 2 ? ncut : 0;
        count = which == 1 ? ncut : which == 2 ? n - ncut : n;
    };
    int64_t efirst, ecount, nfirst, ncount;
    range(mesh.nsharededges, mesh.ncut_edge_groups, efirst, ecount);
    range(mesh.nsharednodes, mesh.ncut_node_groups, nfirst, ncount);
    if (lv.nei == 0) ecount = 0;
    const int64_t total = ecount * lv.nei + ncount;
    if (total > 0) {
        int64_t blocks = (total + 255) / 256;
        if (blocks > cap * 4) blocks = cap * 4;
        hipLaunchKernelGGL(k_iface_csr, dim3((unsigned)blocks), dim3(256), 0, L.stream, mesh.edge_ptr + efirst,
                           mesh.edge_ent, ecount, lv.nei, lv.off_edge, mesh.node_ptr + nfirst, mesh.node_ent, ncount,
                           lv.ld, x);
        check_launch();
    }
}

// ---------------------------------------------------------------------------------------------
// entity masks (Dirichlet constraint / duplicate zeroing)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void entity_range(const LevelDev &lv, int bit, int &off, int &len)
{
    if (bit < lv.nface) {
        off = lv.off_face + bit * lv.nfi;
        len = lv.nfi;
    } else if (bit < lv.nface + lv.nedge) {
        off = lv.off_edge + (bit - lv.nface) * lv.nei;
        len = lv.nei;
    } else {
        off = bit - lv.nface - lv.nedge;
        len = 1;
    }
}

__global__ void __launch_bounds__(64) k_mask(LevelDev lv, int64_t ncells, const uint16_t *__restrict__ mask, double *x)
{
    const int64_t cell = blockIdx.x;
    if (cell >= ncells) return;
    uint32_t m = mask[cell];
    double *xc = x + cell * lv.ld;
    while (m) {
        const int bit = __ffs((int)m) - 1;
        m &= m - 1;
        int off, len;
        entity_range(lv, bit, off, len);
        for (int k = threadIdx.x; k < len; k += 64) xc[off + k] = 0.0;
    }
}

void launch_mask(const Launch &L, const LevelDev &lv, const MeshDev &mesh, double *x, int which)
{
    hipLaunchKernelGGL(k_mask, dim3((unsigned)mesh.ncells), dim3(64), 0, L.stream, lv, mesh.ncells,
                       which == 0 ? mesh.dmask : mesh.dupmask, x);
    check_launch();
}

// ---------------------------------------------------------------------------------------------
// level transfer
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(NT)
k_prolong_add(LevelDev fine, int nfc, int ldc, const double *__restrict__ xc, double *xf)
{
    extern __shared__ double smem[];
    const int64_t cell = blockIdx.x;
    const double *c = xc + cell * ldc;
    for (int t = threadIdx.x; t < nfc; t += NT) smem[t] = c[t];
    __syncthreads();
    double *f = xf + cell * fine.ld;
    for (int t = threadIdx.x; t < fine.nf; t += NT) {
        const int a = fine.par_a[t], b = fine.par_b[t];
        double y = f[t];
        if (a == b)
            y += 1.0 * smem[a];
        else {
            y += 0.5 * smem[a];   // CSC column order: parent with the smaller hierarchical id first
            y += 0.5 * smem[b];
        }
        f[t] = y;
    }
}

// The same for cells of thousands of nodes (level 7: 47 905 fine, 6 545 coarse nodes per cell -- the prolongation of the
// slab levels is a pass of its own): 1024 threads, two workgroups per CU, SPT column entries and packed parent words per
// thread in flight before the first use (the one-entry-per-trip loop above ran at 3.8 TB/s there).  Same roundings.
template <int NT, int SPT>
__global__ void __launch_bounds__(NT)
k_prolong_add_big(LevelDev fine, int nfc, int ldc, const double *__restrict__ xc, double *xf)
{
    extern __shared__ double smem[];
    const int64_t cell = blockIdx.x;
    const double *c = xc + cell * ldc;
    for (int t0 = threadIdx.x; t0 < nfc; t0 += NT * SPT) {
        double v[SPT];
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int t = t0 + q * NT;
            v[q] = t < nfc ? c[t] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int t = t0 + q * NT;
            if (t < nfc) smem[t] = v[q];
        }
    }
    __syncthreads();
    double *f = xf + cell * fine.ld;
    for (int t0 = threadIdx.x; t0 < fine.nf; t0 += NT * SPT) {
        double y[SPT];
        uint32_t w[SPT];
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int t = t0 + q * NT;
            if (t < fine.nf) {
                y[q] = f[t];
                w[q] = fine.par32[t];
            }
        }
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int t = t0 + q * NT;
            if (t < fine.nf) {
                const uint32_t a = w[q] & 0xffffu, b = w[q] >> 16;
                double r = y[q];
                if (a == b)
                    r += 1.0 * smem[a];
                else {
                    r += 0.5 * smem[a];   // CSC column order: parent with the smaller hierarchical id first
                    r += 0.5 * smem[b];
                }
                f[t] = r;
            }
        }
    }
}

template <int NT, bool USE_LDS>
__global__ void __launch_bounds__(NT)
k_restrict(LevelDev fine, int nfc, int ldc, const double *__restrict__ rf, double *bc)
{
    extern __shared__ double smem[];
    const int64_t cell = blockIdx.x;
    const double *f = rf + cell * fine.ld;
    const double *fs = f;
    if (USE_LDS) {
        constexpr int SPT = 8;   // loads first, LDS writes second: keeps 8 loads per thread in flight
        for (int t0 = threadIdx.x; t0 < fine.nf; t0 += NT * SPT) {
            double v[SPT];
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int t = t0 + q * NT;
                v[q] = t < fine.nf ? f[t] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int t = t0 + q * NT;
                if (t < fine.nf) smem[t] = v[q];
            }
        }
        __syncthreads();
        fs = smem;
    }
    double *c = bc + cell * ldc;
    for (int t = threadIdx.x; t < nfc; t += NT) {
        const int b = fine.rptr[t], n = fine.rptr[t + 1] - b;   // 1 <= n <= 15 (3D) / 7 (2D)
        int idx[15];
#pragma unroll
        for (int q = 0; q < 15; ++q) idx[q] = fine.ridx[b + (q < n ? q : 0)];   // all index loads in flight at once
        double tmp = 0.0;
        tmp += 1.0 * fs[idx[0]];
#pragma unroll
        for (int q = 1; q < 15; ++q)
            if (q < n) tmp += 0.5 * fs[idx[q]];
        c[t] = tmp;
    }
}

void launch_prolong_add(const Launch &L, const LevelDev &fine, const LevelDev &coarse, int64_t ncells,
                        const double *xc, double *xf)
{
    size_t lds = sizeof(double) * coarse.nf;
    if (lds > 160 * 1024) throw std::runtime_error("prolongation: coarse cell does not fit LDS");
    if (fine.nf <= 256) {
        auto k = k_prolong_add<64>;
        hipLaunchKernelGGL(k, dim3((unsigned)ncells), dim3(64), lds, L.stream, fine, coarse.nf, coarse.ld, xc, xf);
    } else if (fine.nf > 8192) {
        auto k = k_prolong_add_big<1024, 4>;
        if (lds > 48 * 1024)
            HMG_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3((unsigned)ncells), dim3(1024), lds, L.stream, fine, coarse.nf, coarse.ld, xc, xf);
    } else {
        auto k = k_prolong_add<256>;
        if (lds > 48 * 1024)
            HMG_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3((unsigned)ncells), dim3(256), lds, L.stream, fine, coarse.nf, coarse.ld, xc, xf);
    }
    check_launch();
}

// Restriction of a level whose cell exceeds the LDS: b_coarse[c] = r[c] + 0.5 * sum of r over the fine lattice
// neighbours of c that exist in the cell -- a cell-independent 15-point stencil evaluated at the even fine nodes,
// run through k_apply_slab's rolling window (fine.ctab -> restriction weights, st -> lists of the even nodes with
// their COARSE storage slots).
void launch_restrict_slab(const Launch &L, const LevelDev &fine_rtab, const MeshDev &mesh, const SlabTables &st,
                          int ldc, const double *rf, double *bc)
{
    ApplyArgs a{};
    a.alpha = 1.0;
    a.x = rf;
    a.out = bc;
    a.out_ld = ldc;
    a.flags = 4;
    {   // round 5: through the role-split window kernel with EIGHT loader waves -- an eighth of the nodes is evaluated (the even ones), so
        // the loaders set the pace (option restrict_slab2 = 0: k_apply_slab as in rounds 1-4; the coarse vector is the same to the last bit)
        MeshDev m2 = mesh;
        m2.slab = st;
        if (L.restrict_slab2 && apply_slab2_ok(L, fine_rtab, m2, a)) {
            launch_apply_slab2(L, fine_rtab, m2, a, false);
            return;
        }
    }
    auto kern = k_apply_slab<3, 1024, false>;
    const size_t bytes = sizeof(double) * (size_t)(WSZ + st.lds_nodes);
    HMG_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)mesh.ncells), dim3(1024), bytes, L.stream, fine_rtab, mesh.coef, mesh.dmask, a, st);
    check_launch();
}

void launch_restrict(const Launch &L, const LevelDev &fine, const LevelDev &coarse, int64_t ncells,
                     const double *rf, double *bc)
{
    size_t lds = sizeof(double) * fine.nf;
    if (fine.nf <= 256) {
        hipLaunchKernelGGL((k_restrict<64, true>), dim3((unsigned)ncells), dim3(64), lds, L.stream, fine, coarse.nf,
                           coarse.ld, rf, bc);
    } else if (lds <= 160 * 1024 && fine.nf > 2048) {
        auto k = k_restrict<1024, true>;
        if (lds > 48 * 1024)
            HMG_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3((unsigned)ncells), dim3(1024), lds, L.stream, fine, coarse.nf, coarse.ld, rf, bc);
    } else if (lds <= 160 * 1024) {
        auto k = k_restrict<256, true>;
        if (lds > 48 * 1024)
            HMG_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3((unsigned)ncells), dim3(256), lds, L.stream, fine, coarse.nf, coarse.ld, rf, bc);
    } else {
        hipLaunchKernelGGL((k_restrict<256, false>), dim3((unsigned)ncells), dim3(256), 0, L.stream, fine, coarse.nf,
                           coarse.ld, rf, bc);
    }
    check_launch();
}

// ---------------------------------------------------------------------------------------------
// streaming vector kernels (flat storage, every copy of a shared DOF counted -- BLAS semantics)
//
// Launch shape (profiles/r02_stream_variants.txt): ONE double2 per thread, 256-thread blocks, as many blocks as
// there are pairs -- no grid-stride loop.  The dispatcher hands out blocks in order, so the ~2000 resident blocks
// sweep one contiguous 8 MB window through every stream; a persistent grid-stride grid touches addresses one
// grid-stride (8 MB) apart instead and loses 15 % (5.1 vs 6.0 TB/s on the 24 B/DOF update, 5.1 vs 6.3 TB/s on a
// copy).  Reductions leave one partial per block in L.rpart (deterministic: fixed block order, two more stages).
// ---------------------------------------------------------------------------------------------
constexpr int SB = 256;   // threads per streaming block

// grid of the kernels that keep a grid-stride loop (index arithmetic per element: permutation, synthetic fill, ...)
static inline int strided_blocks(const Launch &L, int64_t n, int per_thread)
{
    int64_t b = (n + (int64_t)256 * per_thread - 1) / ((int64_t)256 * per_thread);
    int64_t cap = (int64_t)L.num_cu * 8;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

static inline int64_t stream_blocks(int64_t n)
{
    const int64_t b = ((n >> 1) + SB - 1) / SB;
    return b < 1 ? 1 : b;
}

static void check_grid(int64_t nb)
{
    if (nb > 0x7fffffffLL) throw std::runtime_error("vector too long for one launch");
}

// block partials [nb] -> 256 partials in `out` (block b sums a contiguous range: fixed order)
__global__ void __launch_bounds__(256) k_reduce_partials(const double *__restrict__ part, int64_t nb, double *out)
{
    __shared__ double red[4];
    const int64_t per = (nb + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < nb ? b0 + per : nb;
    double a = 0.0;
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) a += part[i];
    const double s = block_sum(a, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

__global__ void __launch_bounds__(SB) k_fill(double *x, int64_t n, double v)
{
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i < (n >> 1)) reinterpret_cast<double2 *>(x)[i] = make_double2(v, v);
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) x[n - 1] = v;
}

__global__ void __launch_bounds__(SB) k_copy(double *dst, const double *__restrict__ src, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i < (n >> 1)) reinterpret_cast<double2 *>(dst)[i] = reinterpret_cast<const double2 *>(src)[i];
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
}

__global__ void __launch_bounds__(SB) k_axpy(double a, const double *__restrict__ x, double *y, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i < (n >> 1)) {
        double2 *y2 = reinterpret_cast<double2 *>(y);
        const double2 xv = reinterpret_cast<const double2 *>(x)[i];
        double2 yv = y2[i];
        yv.x = axpy1(a, xv.x, yv.x);
        yv.y = axpy1(a, xv.y, yv.y);
        y2[i] = yv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = axpy1(a, x[n - 1], y[n - 1]);
}

__global__ void __launch_bounds__(SB) k_xpby(const double *__restrict__ r, double b, double *p, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i < (n >> 1)) {
        double2 *p2 = reinterpret_cast<double2 *>(p);
        const double2 rv = reinterpret_cast<const double2 *>(r)[i];
        double2 pv = p2[i];
        pv.x = axpy1(b, pv.x, rv.x);
        pv.y = axpy1(b, pv.y, rv.y);
        p2[i] = pv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = axpy1(b, p[n - 1], r[n - 1]);
}

__global__ void __launch_bounds__(SB)
k_dot(const double *__restrict__ x, const double *__restrict__ y, int64_t n, double *partials)
{
    __shared__ double red[4];
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    double acc = 0.0;
    if (i < (n >> 1)) {
        const double2 xv = reinterpret_cast<const double2 *>(x)[i], yv = reinterpret_cast<const double2 *>(y)[i];
        acc += xv.x * yv.x;
        acc += xv.y * yv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc += x[n - 1] * y[n - 1];
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ void __launch_bounds__(SB)
k_copy_dot(double *p, const double *__restrict__ r, int64_t n, double *partials)
{
    __shared__ double red[4];
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    double acc = 0.0;
    if (i < (n >> 1)) {
        const double2 rv = reinterpret_cast<const double2 *>(r)[i];
        reinterpret_cast<double2 *>(p)[i] = rv;
        acc += rv.x * rv.x;
        acc += rv.y * rv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        p[n - 1] = r[n - 1];
        acc += r[n - 1] * r[n - 1];
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ void __launch_bounds__(SB)
k_cg_update(double *x, double *r, const double *__restrict__ p, const double *__restrict__ q, int64_t n,
            const double *__restrict__ scal, int s_num, int s_den, double *partials)
{
    __shared__ double red[4];
    const double alpha = scal[s_num] / scal[s_den];
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    double acc = 0.0;
    if (i < (n >> 1)) {
        double2 *x2 = reinterpret_cast<double2 *>(x);
        double2 *r2 = reinterpret_cast<double2 *>(r);
        double2 xv = x2[i], rv = r2[i];
        const double2 pv = reinterpret_cast<const double2 *>(p)[i], qv = reinterpret_cast<const double2 *>(q)[i];
        xv.x = axpy1(alpha, pv.x, xv.x);
        xv.y = axpy1(alpha, pv.y, xv.y);
        rv.x = axpy1(-alpha, qv.x, rv.x);
        rv.y = axpy1(-alpha, qv.y, rv.y);
        x2[i] = xv;
        r2[i] = rv;
        acc += rv.x * rv.x;
        acc += rv.y * rv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        x[n - 1] = axpy1(alpha, p[n - 1], x[n - 1]);
        double rv = axpy1(-alpha, q[n - 1], r[n - 1]);
        r[n - 1] = rv;
        acc += rv * rv;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// rout = r - alpha q, partial rout.rout   (alpha = scal[s_num] / scal[s_den]; rout may be r); the x-update rides with
// the next fused apply
__global__ void __launch_bounds__(SB)
k_cg_rupdate(const double *r, double *rout, const double *__restrict__ q, int64_t n, const double *__restrict__ scal,
             int s_num, int s_den, double *partials)
{
    __shared__ double red[4];
    const double alpha = scal[s_num] / scal[s_den];
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    double acc = 0.0;
    if (i < (n >> 1)) {
        double2 rv = reinterpret_cast<const double2 *>(r)[i];
        const double2 qv = reinterpret_cast<const double2 *>(q)[i];
        rv.x = axpy1(-alpha, qv.x, rv.x);
        rv.y = axpy1(-alpha, qv.y, rv.y);
        reinterpret_cast<double2 *>(rout)[i] = rv;
        acc += rv.x * rv.x;
        acc += rv.y * rv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double rv = axpy1(-alpha, q[n - 1], r[n - 1]);
        rout[n - 1] = rv;
        acc += rv * rv;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// k_cg_rupdate with the FACE part of q's interface sum done on the fly: q holds cell-local values on the shared faces
// (edges, corners and partition-cut entities are already summed in place), the partner copy is fetched through the
// per-cell table fpart[4 * cell + face] = partner cell << 2 | partner face (-1: none) and added in ascending cell order --
// the bits of the separate pass (k_iface_faces + k_cg_rupdate).  Saves that pass's read-modify-write of 28 % of q
// (16 B per face DOF) for 8 B per face DOF more here.  q itself stays unsummed on the faces.
// XU > 0 (the last step of a smoother whose caller reads x and r only -- the finest level's post-smoother inside hmg_vcycle, option
// lazy_top): the apply of that step formed p_i = r_i + beta p_{i-1} in LDS only and wrote neither p nor x; both pending x-updates
// are done here, p_i formed again on the fly from the r this pass reads anyway:  x = (x + ax p) + alpha (r + beta p)  -- the
// roundings of  x += alpha_{i-1} p_{i-1};  p_i = r_i + beta p_{i-1};  x += alpha_i p_i  one after the other (k_cg_x2_update).
// XU == 2: a third pending update comes first, x += az p0 (the step before wrote its direction next to p0 instead of over it).
struct XUSlots {
    int a_num, a_den, b_num, b_den, z_num, z_den;
};
template <int XU>
__global__ void __launch_bounds__(SB)
k_cg_rupdate_faces(const double *r, double *rout, const double *__restrict__ q, int64_t n, const double *__restrict__ scal,
                   int s_num, int s_den, double *partials, const int32_t *__restrict__ fpart, int ld, double inv_ld,
                   int off_face, int off_int, int nfi, double *x, const double *__restrict__ p, const double *__restrict__ p0,
                   XUSlots xs)
{
    __shared__ double red[4];
    const double alpha = scal[s_num] / scal[s_den];
    const double ax = XU ? scal[xs.a_num] / scal[xs.a_den] : 0.0;
    const double beta = XU ? scal[xs.b_num] / scal[xs.b_den] : 0.0;
    const double az = XU == 2 ? scal[xs.z_num] / scal[xs.z_den] : 0.0;
    auto xupd = [&](double xv, double pv, double rv) {
        const double t1 = axpy1(ax, pv, xv);
        const double p2 = axpy1(beta, pv, rv);
        return axpy1(alpha, p2, t1);
    };
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    double acc = 0.0;
    auto summed = [&](int64_t e, double own) {          // q[e] with its face partner, if it has one
        int64_t cell = (int64_t)((double)e * inv_ld);
        int64_t slot = e - cell * ld;
        if (slot < 0) {
            cell -= 1;
            slot += ld;
        } else if (slot >= ld) {
            cell += 1;
            slot -= ld;
        }
        if (slot < off_face || slot >= off_int) return own;
        const int s = (int)slot - off_face;
        const int f = (s >= nfi) + (s >= 2 * nfi) + (s >= 3 * nfi);
        const int32_t p = fpart[4 * cell + f];
        if (p < 0) return own;
        const int64_t pc = p >> 2;
        const double other = q[pc * ld + off_face + (p & 3) * nfi + (s - f * nfi)];
        return cell < pc ? own + other : other + own;
    };
    if (i < (n >> 1)) {
        double2 rv = reinterpret_cast<const double2 *>(r)[i];
        const double2 qv = reinterpret_cast<const double2 *>(q)[i];
        if (XU) {
            double2 xv = reinterpret_cast<double2 *>(x)[i];
            const double2 pv = reinterpret_cast<const double2 *>(p)[i];
            if (XU == 2) {
                const double2 zv = reinterpret_cast<const double2 *>(p0)[i];
                xv.x = axpy1(az, zv.x, xv.x);
                xv.y = axpy1(az, zv.y, xv.y);
            }
            xv.x = xupd(xv.x, pv.x, rv.x);
            xv.y = xupd(xv.y, pv.y, rv.y);
            reinterpret_cast<double2 *>(x)[i] = xv;
        }
        rv.x = axpy1(-alpha, summed(2 * i, qv.x), rv.x);
        rv.y = axpy1(-alpha, summed(2 * i + 1, qv.y), rv.y);
        reinterpret_cast<double2 *>(rout)[i] = rv;
        acc += rv.x * rv.x;
        acc += rv.y * rv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        if (XU) x[n - 1] = xupd(XU == 2 ? axpy1(az, p0[n - 1], x[n - 1]) : x[n - 1], p[n - 1], r[n - 1]);
        double rv = axpy1(-alpha, summed(n - 1, q[n - 1]), r[n - 1]);
        rout[n - 1] = rv;
        acc += rv * rv;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// x += alpha p (alpha = scal[a_num]/scal[a_den]) and, if with_p, p = r + beta p (beta = scal[s_num]/scal[s_den])
__global__ void __launch_bounds__(SB)
k_cg_xp_update(double *x, double *p, const double *__restrict__ r, int64_t n, const double *__restrict__ scal,
               int a_num, int a_den, int s_num, int s_den, int with_p)
{
    const double alpha = scal[a_num] / scal[a_den];
    const double beta = with_p ? scal[s_num] / scal[s_den] : 0.0;
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i < (n >> 1)) {
        double2 *x2 = reinterpret_cast<double2 *>(x);
        double2 *p2 = reinterpret_cast<double2 *>(p);
        double2 xv = x2[i], pv = p2[i];
        double2 rv = make_double2(0.0, 0.0);
        if (with_p) rv = reinterpret_cast<const double2 *>(r)[i];
        xv.x = axpy1(alpha, pv.x, xv.x);
        xv.y = axpy1(alpha, pv.y, xv.y);
        x2[i] = xv;
        if (with_p) {
            pv.x = axpy1(beta, pv.x, rv.x);
            pv.y = axpy1(beta, pv.y, rv.y);
            p2[i] = pv;
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        x[n - 1] = axpy1(alpha, p[n - 1], x[n - 1]);
        if (with_p) p[n - 1] = axpy1(beta, p[n - 1], r[n - 1]);
    }
}

// two CG x-updates at once, the second direction formed on the fly (a post-smoother's dead last step below the finest level,
// see smooth()): x = (x + (scal[a_num]/scal[a_den]) p) + (scal[c_num]/scal[c_den]) (r + (scal[b_num]/scal[b_den]) p) -- the three
// roundings of x += alpha_0 p_0; p_1 = r + beta p_0; x += alpha_1 p_1 done one after the other (k_apply's x3 mode does the same)
__global__ void __launch_bounds__(SB)
k_cg_x2_update(double *x, const double *__restrict__ p, const double *__restrict__ r, int64_t n, const double *__restrict__ scal,
               int a_num, int a_den, int b_num, int b_den, int c_num, int c_den)
{
    const double ax = scal[a_num] / scal[a_den];
    const double beta = scal[b_num] / scal[b_den];
    const double c2 = scal[c_num] / scal[c_den];
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    auto upd = [&](double xv, double pv, double rv) {
        const double t1 = axpy1(ax, pv, xv);
        const double p2 = axpy1(beta, pv, rv);
        return axpy1(c2, p2, t1);
    };
    if (i < (n >> 1)) {
        double2 *x2 = reinterpret_cast<double2 *>(x);
        double2 xv = x2[i];
        const double2 pv = reinterpret_cast<const double2 *>(p)[i], rv = reinterpret_cast<const double2 *>(r)[i];
        xv.x = upd(xv.x, pv.x, rv.x);
        xv.y = upd(xv.y, pv.y, rv.y);
        x2[i] = xv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) x[n - 1] = upd(x[n - 1], p[n - 1], r[n - 1]);
}

__global__ void __launch_bounds__(SB)
k_cg_pupdate(double *p, const double *__restrict__ r, int64_t n, const double *__restrict__ scal, int s_num, int s_den)
{
    const double beta = scal[s_num] / scal[s_den];
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i < (n >> 1)) {
        double2 *p2 = reinterpret_cast<double2 *>(p);
        const double2 rv = reinterpret_cast<const double2 *>(r)[i];
        double2 pv = p2[i];
        pv.x = axpy1(beta, pv.x, rv.x);
        pv.y = axpy1(beta, pv.y, rv.y);
        p2[i] = pv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = axpy1(beta, p[n - 1], r[n - 1]);
}

static void finalize(const Launch &L, int nb, int slot)
{
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, L.stream, L.partials, nb, L.scal, slot);
    check_launch();
}

// Where a streaming reduction over nb blocks leaves its partials, and how they become scal[slot]: few blocks go to the
// small buffer and one finalize; many go to L.rpart, are folded to 256 partials, then finalized.
static double *reduce_target(const Launch &L, int64_t nb)
{
    if (nb <= 2048) return L.partials;
    if (!L.rpart || nb > L.rpart_cap) throw std::runtime_error("reduction scratch too small for this vector");
    return L.rpart;
}

static void reduce_finish(const Launch &L, int64_t nb, int slot)
{
    if (nb <= 2048) {
        finalize(L, (int)nb, slot);
        return;
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3(256), dim3(256), 0, L.stream, L.rpart, nb, L.partials);
    check_launch();
    finalize(L, 256, slot);
}

void launch_fill(const Launch &L, double *x, int64_t n, double v)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)nb), dim3(SB), 0, L.stream, x, n, v);
    check_launch();
}
void launch_copy(const Launch &L, double *dst, const double *src, int64_t n)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    hipLaunchKernelGGL(k_copy, dim3((unsigned)nb), dim3(SB), 0, L.stream, dst, src, n);
    check_launch();
}
void launch_axpy(const Launch &L, double a, const double *x, double *y, int64_t n)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    hipLaunchKernelGGL(k_axpy, dim3((unsigned)nb), dim3(SB), 0, L.stream, a, x, y, n);
    check_launch();
}
void launch_xpby(const Launch &L, const double *r, double b, double *p, int64_t n)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    hipLaunchKernelGGL(k_xpby, dim3((unsigned)nb), dim3(SB), 0, L.stream, r, b, p, n);
    check_launch();
}
void launch_dot(const Launch &L, const double *x, const double *y, int64_t n, int slot)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    hipLaunchKernelGGL(k_dot, dim3((unsigned)nb), dim3(SB), 0, L.stream, x, y, n, reduce_target(L, nb));
    check_launch();
    reduce_finish(L, nb, slot);
}
void launch_copy_dot(const Launch &L, double *p, const double *r, int64_t n, int slot)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    hipLaunchKernelGGL(k_copy_dot, dim3((unsigned)nb), dim3(SB), 0, L.stream, p, r, n, reduce_target(L, nb));
    check_launch();
    reduce_finish(L, nb, slot);
}
void launch_cg_update(const Launch &L, double *x, double *r, const double *p, const double *q, int64_t n, int s_num,
                      int s_den, int s_out)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    hipLaunchKernelGGL(k_cg_update, dim3((unsigned)nb), dim3(SB), 0, L.stream, x, r, p, q, n, L.scal, s_num, s_den,
                       reduce_target(L, nb));
    check_launch();
    reduce_finish(L, nb, s_out);
}
void launch_cg_rupdate(const Launch &L, const double *r, double *rout, const double *q, int64_t n, int s_num, int s_den,
                       int s_out)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    hipLaunchKernelGGL(k_cg_rupdate, dim3((unsigned)nb), dim3(SB), 0, L.stream, r, rout, q, n, L.scal, s_num, s_den,
                       reduce_target(L, nb));
    check_launch();
    reduce_finish(L, nb, s_out);
}

void launch_cg_rupdate_faces(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const double *r, double *rout,
                             const double *q, int64_t n, int s_num, int s_den, int s_out)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    hipLaunchKernelGGL(k_cg_rupdate_faces<0>, dim3((unsigned)nb), dim3(SB), 0, L.stream, r, rout, q, n, L.scal, s_num, s_den,
                       reduce_target(L, nb), mesh.face_partner, lv.ld, 1.0 / (double)lv.ld, lv.off_face, lv.off_int, lv.nfi,
                       (double *)nullptr, (const double *)nullptr, (const double *)nullptr, XUSlots{0, 0, 0, 0, 0, 0});
    check_launch();
    reduce_finish(L, nb, s_out);
}

// ... with the pending x-updates of a lazy last step: x = ((x [+ (z_num/z_den) p0]) + (a_num/a_den) p) + (s_num/s_den) (r + (b_num/b_den) p)
void launch_cg_rupdate_faces_x(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const double *r, double *rout,
                               const double *q, int64_t n, int s_num, int s_den, int s_out, double *x, const double *p, int a_num,
                               int a_den, int b_num, int b_den, const double *p0, int z_num, int z_den)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    const XUSlots xs{a_num, a_den, b_num, b_den, z_num, z_den};
    if (p0)
        hipLaunchKernelGGL(k_cg_rupdate_faces<2>, dim3((unsigned)nb), dim3(SB), 0, L.stream, r, rout, q, n, L.scal, s_num, s_den,
                           reduce_target(L, nb), mesh.face_partner, lv.ld, 1.0 / (double)lv.ld, lv.off_face, lv.off_int, lv.nfi, x, p,
                           p0, xs);
    else
        hipLaunchKernelGGL(k_cg_rupdate_faces<1>, dim3((unsigned)nb), dim3(SB), 0, L.stream, r, rout, q, n, L.scal, s_num, s_den,
                           reduce_target(L, nb), mesh.face_partner, lv.ld, 1.0 / (double)lv.ld, lv.off_face, lv.off_int, lv.nfi, x, p,
                           p0, xs);
    check_launch();
    reduce_finish(L, nb, s_out);
}

void launch_cg_xp_update(const Launch &L, double *x, double *p, const double *r, int64_t n, int a_num, int a_den,
                         int s_num, int s_den, int with_p)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    hipLaunchKernelGGL(k_cg_xp_update, dim3((unsigned)nb), dim3(SB), 0, L.stream, x, p, r, n, L.scal, a_num, a_den, s_num,
                       s_den, with_p);
    check_launch();
}

void launch_cg_x2_update(const Launch &L, double *x, const double *p, const double *r, int64_t n, int a_num, int a_den, int b_num,
                         int b_den, int c_num, int c_den)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    hipLaunchKernelGGL(k_cg_x2_update, dim3((unsigned)nb), dim3(SB), 0, L.stream, x, p, r, n, L.scal, a_num, a_den, b_num, b_den,
                       c_num, c_den);
    check_launch();
}

void launch_cg_pupdate(const Launch &L, double *p, const double *r, int64_t n, int s_num, int s_den)
{
    const int64_t nb = stream_blocks(n);
    check_grid(nb);
    hipLaunchKernelGGL(k_cg_pupdate, dim3((unsigned)nb), dim3(SB), 0, L.stream, p, r, n, L.scal, s_num, s_den);
    check_launch();
}

// sum over first copies only: cell-wise, entities flagged in dupmask skipped
__global__ void __launch_bounds__(256)
k_norm2_unique(LevelDev lv, int64_t ncells, const uint16_t *__restrict__ dupmask, const double *__restrict__ x,
               double *partials)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t cell = blockIdx.x; cell < ncells; cell += gridDim.x) {
        const uint32_t dm = dupmask[cell];
        const double *xc = x + cell * lv.ld;
        for (int t = threadIdx.x; t < lv.nf; t += 256) {
            const int cls = (int)((lv.meta[t] >> 24) & 0xffu);
            const bool dup = cls > 0 && ((dm >> (cls - 1)) & 1u);
            const double v = xc[t];
            if (!dup) acc += v * v;
        }
    }
    double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

void launch_norm2_unique(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const double *x, int slot)
{
    int64_t nb = mesh.ncells;
    if (nb > (int64_t)L.num_cu * 8) nb = (int64_t)L.num_cu * 8;
    hipLaunchKernelGGL(k_norm2_unique, dim3((unsigned)nb), dim3(256), 0, L.stream, lv, mesh.ncells, mesh.dupmask, x,
                       L.partials);
    check_launch();
    finalize(L, (int)nb, slot);
}

// ---------------------------------------------------------------------------------------------
// level-1 gather / scatter (level-1 storage slot == local node id)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_gather_base(const int32_t *__restrict__ node_first, int64_t nnodes, int ld1, const double *__restrict__ v1, double *u)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nnodes) return;
    const int32_t v = node_first[g];
    u[g] = v >= 0 ? v1[(int64_t)(v >> 3) * ld1 + (v & 7)] : 0.0;
}

__global__ void __launch_bounds__(256)
k_scatter_base(const int32_t *__restrict__ cells, int64_t ncells, int npc, int ld1, const double *__restrict__ u,
               double *v1)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncells * npc) return;
    const int64_t c = i / npc;
    const int l = (int)(i - c * npc);
    v1[c * ld1 + l] = u[cells[i]];
}

// partitioned gather: u_global[nodes_g[q]] = v1[first local copy of local node q], owned nodes only
__global__ void __launch_bounds__(256)
k_gather_owned(const int32_t *__restrict__ node_first, const int32_t *__restrict__ nodes_g,
               const int32_t *__restrict__ owned, int64_t nlocal, int ld1, const double *__restrict__ v1, double *ug)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nlocal || !owned[q]) return;
    const int32_t v = node_first[q];
    ug[nodes_g[q]] = v1[(int64_t)(v >> 3) * ld1 + (v & 7)];
}

void launch_gather_owned(const Launch &L, const MeshDev &mesh, const int32_t *nodes_g, const int32_t *owned, int ld1,
                         const double *v1, double *ug)
{
    hipLaunchKernelGGL(k_gather_owned, dim3((unsigned)((mesh.nnodes + 255) / 256)), dim3(256), 0, L.stream,
                       mesh.node_first, nodes_g, owned, mesh.nnodes, ld1, v1, ug);
    check_launch();
}

void launch_scatter_cells(const Launch &L, const int32_t *cell_nodes, int64_t ncells, int npc, int ld1, const double *u,
                          double *v1)
{
    hipLaunchKernelGGL(k_scatter_base, dim3((unsigned)((ncells * npc + 255) / 256)), dim3(256), 0, L.stream,
                       cell_nodes, ncells, npc, ld1, u, v1);
    check_launch();
}

void launch_gather_base(const Launch &L, const MeshDev &mesh, int ld1, const double *v1, double *u)
{
    hipLaunchKernelGGL(k_gather_base, dim3((unsigned)((mesh.nnodes + 255) / 256)), dim3(256), 0, L.stream,
                       mesh.node_first, mesh.nnodes, ld1, v1, u);
    check_launch();
}
void launch_scatter_base(const Launch &L, const MeshDev &mesh, int ld1, const double *u, double *v1)
{
    const int npc = mesh.dim + 1;
    hipLaunchKernelGGL(k_scatter_base, dim3((unsigned)((mesh.ncells * npc + 255) / 256)), dim3(256), 0, L.stream,
                       mesh.cells, mesh.ncells, npc, ld1, u, v1);
    check_launch();
}

// ---------------------------------------------------------------------------------------------
// API order (hierarchical, Nf x Ne column-major) <-> storage order; deterministic synthetic fill
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_permute(LevelDev lv, int64_t ncells, const double *__restrict__ src, double *dst, int to_storage)
{
    const int64_t total = ncells * lv.nf;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i / lv.nf;
        const int h = (int)(i - c * lv.nf);
        const int s = lv.hier2slot[h];
        if (to_storage)
            dst[c * lv.ld + s] = src[i];
        else
            dst[i] = src[c * lv.ld + s];
    }
}

void launch_permute(const Launch &L, const LevelDev &lv, int64_t ncells, const double *src, double *dst, int to_storage)
{
    hipLaunchKernelGGL(k_permute, dim3(strided_blocks(L, ncells * lv.nf, 4)), dim3(256), 0, L.stream, lv, ncells, src,
                       dst, to_storage);
    check_launch();
}

__device__ __forceinline__ double hash_u01(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed + (idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// x[hier h, cell c] = u01(seed, ((c + cell_offset) << 20) | h): independent of the storage order
__global__ void __launch_bounds__(256)
k_fill_random(LevelDev lv, int64_t ncells, double *x, uint64_t seed, int64_t cell_offset)
{
    const int64_t total = ncells * lv.nf;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i / lv.nf;
        const int h = (int)(i - c * lv.nf);
        x[c * lv.ld + lv.hier2slot[h]] = hash_u01(seed, ((uint64_t)(c + cell_offset) << 20) | (uint64_t)h);
    }
}

void launch_fill_random(const Launch &L, const LevelDev &lv, int64_t ncells, double *x, uint64_t seed,
                        int64_t cell_offset)
{
    hipLaunchKernelGGL(k_fill_random, dim3(strided_blocks(L, ncells * lv.nf, 4)), dim3(256), 0, L.stream, lv, ncells, x,
                       seed, cell_offset);
    check_launch();
}

// ---------------------------------------------------------------------------------------------
// coarse (level-1) preconditioned CG (Jacobi, or Chebyshev iterates of the Jacobi-scaled operator: k_coarse_cheb) on the assembled interior matrix
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_coarse_gather_rhs(const int32_t *__restrict__ interior, int64_t n, const double *__restrict__ u, double *b)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = u[interior[i]];
}

__global__ void __launch_bounds__(256)
k_coarse_scatter_sol(const int32_t *__restrict__ interior, int64_t n, const double *__restrict__ x, double *u)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) u[interior[i]] = x[i];
}

__global__ void __launch_bounds__(256)
k_coarse_init(CoarseDev A, const double *__restrict__ b, double *x, double *r, double *z, double *p, double *part0,
              double *part1, double zscale, double *dcheb)
{
    // zscale, dcheb: polynomial preconditioner (k_coarse_cheb) -- z = d_0 = (1 / theta) D^-1 r is the first Chebyshev iterate
    __shared__ double red[4];
    double rz = 0.0, bb = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * blockDim.x) {
        const double bi = b[i];
        const double zi = dcheb ? zscale * (bi / A.diag[i]) : bi / A.diag[i];
        x[i] = 0.0;
        r[i] = bi;
        z[i] = zi;
        p[i] = zi;
        if (dcheb) dcheb[i] = zi;
        rz += bi * zi;
        bb += bi * bi;
    }
    double s0 = block_sum(rz, red);
    double s1 = block_sum(bb, red);
    if (threadIdx.x == 0) {
        part0[blockIdx.x] = s0;
        part1[blockIdx.x] = s1;
    }
}

// Sparse product of the level-1 matrix, CR consecutive rows per group of 16 lanes: row r's entries go to the lanes as before (entry
// rowptr[r] + lane, then every 16th: the Freudenthal lattice has <= 15 per row) and are folded in the same xor tree, so a row's sum
// has the bits it always had -- but the (value, column) pairs and the gathers of CR rows are in flight together.  Round 4: with one
// row per group a product was a chain of dependent round trips, 16 rounds per group at 64^3 cubes with four waves per SIMD: 21 us
// for 49 MB.  Every lane of the group returns all CR sums.
constexpr int CR = 4;
__device__ __forceinline__ void coarse_rows_product(const CoarseDev &A, const double *__restrict__ z, int64_t i0, int sub, double (&s)[CR])
{
    int rp[CR + 1];
#pragma unroll
    for (int j = 0; j <= CR; ++j) rp[j] = A.rowptr[i0 + j < A.n ? i0 + j : A.n];
    double v[CR], zz[CR];
#pragma unroll
    for (int j = 0; j < CR; ++j) {
        const int k = rp[j] + sub;
        const bool ok = k < rp[j + 1];
        const int kc = ok ? k : 0;
        v[j] = A.val[kc];
        zz[j] = z[A.colidx[kc]];       // (not ok: entry 0, a valid address; its product is dropped)
        if (!ok) v[j] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < CR; ++j) {
        s[j] = 0.0;
        s[j] += v[j] * zz[j];
        for (int k = rp[j] + sub + 16; k < rp[j + 1]; k += 16) s[j] += A.val[k] * z[A.colidx[k]];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
#pragma unroll
        for (int j = 0; j < CR; ++j) s[j] += __shfl_xor(s[j], o, 16);
    }
}

static_assert(CR == 4, "the row epilogues pick one of four sums");

// Two-launch form of one PCG iteration (round 3): `direction` + `update`.  With q = A p carried by its own recurrence,
//   p' = z + beta p,   q' = A z + beta q      (A p' = A z + beta A p),
// the sparse product reads z, which the previous update left complete, and never a neighbour's p -- so the p-update, the
// product and the partial sums of p.q are one kernel; beta, the convergence test and the bookkeeping that k_coarse_pupdate did
// ride in its head (every block sums the same partials in the same order: the decision is the same in all of them).  On the
// level-1 systems of configs 3-5 the recurrence changes neither the iteration count nor the true residual (7.7e-14 / 9.2e-14
// at contrast 9 / 100, x equal to 5e-16): tools/dev note in profiles/r03_experiments.txt.
//   mode 0: a regular iteration;  1: the first of a solve (p = z and rz come from k_coarse_init: q = A z);
//   2: bookkeeping only (behind the last update of a batch, so that the host finds the flag and the count up to date)
//   count_it = 0: the update before this launch has been counted already (by a mode-2 launch)
__global__ void __launch_bounds__(256)
k_coarse_direction(CoarseDev A, double *p, double *q, const double *__restrict__ z, double *scal, int slot_old, int slot_new,
                   const double *__restrict__ part_rz, const double *__restrict__ part_rr, int nb_upd, double rtol2, int mode,
                   int count_it, double *part_pq, int nb_rz)
{
    __shared__ double red[4];
    __shared__ double bc;
    __shared__ int skip;
    if (threadIdx.x == 0) skip = scal[S_DONE] != 0.0 || !(scal[S_C2] > 0.0);
    __syncthreads();
    if (skip) return;
    double beta = 0.0;
    if (mode != 1) {
        const double rz_new = sum_partials_all(part_rz, nb_rz, red, &bc);   // (polynomial preconditioner: from the last Chebyshev step)
        const double rr = sum_partials_all(part_rr, nb_upd, red, &bc);
        const double rz_old = scal[slot_old];
        beta = rz_old != 0.0 ? rz_new / rz_old : 0.0;
        const bool done = rr <= rtol2 * scal[S_C2];                 // (NaN compares false: the budget runs out, the host sees S_CRR)
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            scal[slot_new] = rz_new;
            if (count_it) scal[S_ITER] += 1.0;
            scal[S_CRR] = rr;
            if (done) scal[S_DONE] = 1.0;
        }
        if (done || mode == 2) return;
    }
    // 16 lanes per row (the level-1 matrix of the Freudenthal lattice has <= 15 entries per row): the lanes of a row read
    // consecutive (value, column) pairs -- coalesced, where one thread per row strides by the row length -- and fold their
    // products in a fixed xor tree, so the result does not depend on the launch shape (64^3 cubes, 274 625 rows: 11.3 -> 4.9 ms
    // per solve in round 2)
    const int sub = threadIdx.x & 15;
    double acc = 0.0;
    for (int64_t i0 = CR * (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4); i0 < A.n; i0 += CR * (((int64_t)gridDim.x * blockDim.x) >> 4)) {
        const int64_t i = i0 + sub;                     // lane j < CR finishes row i0 + j
        const bool mine = sub < CR && i < A.n;
        double zi = 0.0, pi0 = 0.0, qi0 = 0.0;
        if (mine) {
            zi = z[i];
            if (mode != 1) {
                pi0 = p[i];
                qi0 = q[i];
            }
        }
        double sr[CR];
        coarse_rows_product(A, z, i0, sub, sr);
        if (mine) {
            const double s = sub == 0 ? sr[0] : sub == 1 ? sr[1] : sub == 2 ? sr[2] : sr[3];
            const double pi = mode == 1 ? zi : zi + beta * pi0;
            const double qi = mode == 1 ? s : s + beta * qi0;
            p[i] = pi;
            q[i] = qi;
            acc += pi * qi;
        }
    }
    const double sacc = block_sum(acc, red);
    if (threadIdx.x == 0) part_pq[blockIdx.x] = sacc;
}

// One step of the Chebyshev iteration for D^-1 A z = D^-1 r from z_1 = (1 / theta) D^-1 r (Saad, Iterative Methods, Alg. 12.1):
//   d <- c1 d + c2 D^-1 (r - A zin),  zout = zin + d          (c1 = rho_j rho_j-1, c2 = 2 rho_j / delta: host, launch_coarse_cheb)
// k - 1 of these between the update and the direction kernel make z = p_{k-1}(D^-1 A) D^-1 r, a symmetric positive definite
// polynomial preconditioner (the interval [lmax / ratio, lmax] with lmax a Gershgorin bound: the polynomial stays positive on the
// whole spectrum).  No reductions but the last step's partial sums of r.z: an outer CG iteration costs k + 1 launches for k
// sparse products, where plain Jacobi-PCG pays two launches and two grid-wide sums per product.  16 lanes per row as in
// k_coarse_direction.
__global__ void __launch_bounds__(256)
k_coarse_cheb(CoarseDev A, const double *__restrict__ r, const double *__restrict__ zin, double *zout, double *d, double c1, double c2,
              const double *__restrict__ scal, int last, double *part_rz)
{
    __shared__ double red[4];
    if (scal[S_DONE] != 0.0 || !(scal[S_C2] > 0.0)) return;
    const int sub = threadIdx.x & 15;
    double acc = 0.0;
    for (int64_t i0 = CR * (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4); i0 < A.n; i0 += CR * (((int64_t)gridDim.x * blockDim.x) >> 4)) {
        const int64_t i = i0 + sub;                     // lane j < CR finishes row i0 + j
        const bool mine = sub < CR && i < A.n;
        double ri = 0.0, dold = 0.0, dg = 1.0, zold = 0.0;
        if (mine) {
            ri = r[i];
            dold = d[i];
            dg = A.diag[i];
            zold = zin[i];
        }
        double sr[CR];
        coarse_rows_product(A, zin, i0, sub, sr);
        if (mine) {
            const double s = sub == 0 ? sr[0] : sub == 1 ? sr[1] : sub == 2 ? sr[2] : sr[3];
            const double di = c1 * dold + c2 * ((ri - s) / dg);
            const double zi = zold + di;
            d[i] = di;
            zout[i] = zi;
            acc += ri * zi;
        }
    }
    if (last) {
        const double sacc = block_sum(acc, red);
        if (threadIdx.x == 0) part_rz[blockIdx.x] = sacc;
    }
}

// update: sums the partials of p.q itself (P0); partials of r.z -> P1, r.r -> P2.  rz lives in scal[slot_old] /
// scal[slot_new], exchanged by the host every iteration, so no kernel overwrites a scalar that another block of the same
// launch may still read.
__global__ void __launch_bounds__(256)
k_coarse_update(CoarseDev A, double *x, double *r, double *z, const double *__restrict__ p, const double *__restrict__ q,
                const double *__restrict__ scal, int slot_old, const double *__restrict__ part_pap, int nb, double *part_rz,
                double *part_rr, double zscale, double *dcheb)
{
    __shared__ double red[4];
    __shared__ double bc;
    if (scal[S_DONE] != 0.0 || !(scal[S_C2] > 0.0)) return;
    const double pap = sum_partials_all(part_pap, nb, red, &bc);
    // exact convergence (r = 0, e.g. a 1-unknown system after one step) makes p.Ap = 0: stay at the solution
    const double alpha = pap != 0.0 ? scal[slot_old] / pap : 0.0;
    double rz = 0.0, rr = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * blockDim.x) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * q[i];
        const double zi = dcheb ? zscale * (ri / A.diag[i]) : ri / A.diag[i];
        r[i] = ri;
        z[i] = zi;
        if (dcheb) dcheb[i] = zi;
        rz += ri * zi;
        rr += ri * ri;
    }
    double s0 = block_sum(rz, red);
    double s1 = block_sum(rr, red);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = s0;
        part_rr[blockIdx.x] = s1;
    }
}

static inline int coarse_blocks(const Launch &L, int64_t n)
{
    int64_t b = (n + 255) / 256;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

static inline int coarse_spmv_blocks(int64_t n)      // 16 lanes per row
{
    int64_t b = (16 * n + 255) / 256;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

void launch_coarse_gather_rhs(const Launch &L, const CoarseDev &A, const double *u, double *b)
{
    hipLaunchKernelGGL(k_coarse_gather_rhs, dim3((unsigned)((A.n + 255) / 256)), dim3(256), 0, L.stream, A.interior,
                       A.n, u, b);
    check_launch();
}
void launch_coarse_scatter_sol(const Launch &L, const CoarseDev &A, int64_t nnodes, const double *x, double *u)
{
    launch_fill(L, u, nnodes, 0.0);
    hipLaunchKernelGGL(k_coarse_scatter_sol, dim3((unsigned)((A.n + 255) / 256)), dim3(256), 0, L.stream, A.interior,
                       A.n, x, u);
    check_launch();
}
void launch_coarse_init(const Launch &L, const CoarseDev &A, const double *b, double *x, double *r, double *z, double *p,
                        double zscale, double *dcheb)
{
    int nb = coarse_blocks(L, A.n);
    hipLaunchKernelGGL(k_coarse_init, dim3(nb), dim3(256), 0, L.stream, A, b, x, r, z, p, L.partials, L.partials + 2048, zscale, dcheb);
    check_launch();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, L.stream, L.partials, nb, L.scal, (int)S_C0);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, L.stream, L.partials + 2048, nb, L.scal, (int)S_C2);
    HMG_HIP_CHECK(hipMemsetAsync(L.scal + S_DONE, 0, 3 * sizeof(double), L.stream));   // S_DONE, S_ITER, S_CRR
    check_launch();
}

void launch_coarse_cheb(const Launch &L, const CoarseDev &A, const double *r, const double *zin, double *zout, double *d, double c1,
                        double c2, int last)
{
    const int nb = coarse_spmv_blocks(A.n);
    hipLaunchKernelGGL(k_coarse_cheb, dim3(nb), dim3(256), 0, L.stream, A, r, zin, zout, d, c1, c2, L.scal, last, L.partials + 1024);
    check_launch();
}

// scal[S_C0] = r.z from the partials the last Chebyshev step left (first iteration of a solve)
void launch_coarse_rz_from_cheb(const Launch &L, const CoarseDev &A)
{
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, L.stream, L.partials + 1024, coarse_spmv_blocks(A.n), L.scal, (int)S_C0);
    check_launch();
}
// partial buffers of the PCG inside L.partials (>= 4096 doubles, nb <= 1024): P0 p.q, P1 r.z, P2 r.r
void launch_coarse_direction(const Launch &L, const CoarseDev &A, double *p, double *q, const double *z, int slot_old, int slot_new,
                             double rtol2, int mode, int count_it, int rz_from_cheb)
{
    const int nb = mode == 2 ? 1 : coarse_spmv_blocks(A.n);
    hipLaunchKernelGGL(k_coarse_direction, dim3(nb), dim3(256), 0, L.stream, A, p, q, z, L.scal, slot_old, slot_new,
                       L.partials + 1024, L.partials + 2048, coarse_blocks(L, A.n), rtol2, mode, count_it, L.partials,
                       rz_from_cheb ? coarse_spmv_blocks(A.n) : coarse_blocks(L, A.n));
    check_launch();
}

void launch_coarse_update(const Launch &L, const CoarseDev &A, double *x, double *r, double *z, const double *p,
                          const double *q, int slot_old, double zscale, double *dcheb)
{
    int nb = coarse_blocks(L, A.n);
    hipLaunchKernelGGL(k_coarse_update, dim3(nb), dim3(256), 0, L.stream, A, x, r, z, p, q, L.scal, slot_old, L.partials,
                       coarse_spmv_blocks(A.n), L.partials + 1024, L.partials + 2048, zscale, dcheb);
    check_launch();
}
// r.r of the last update -> scal[S_TMP] (convergence check)
void launch_coarse_residual_norm(const Launch &L, const CoarseDev &A)
{
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, L.stream, L.partials + 2048, coarse_blocks(L, A.n), L.scal,
                       (int)S_TMP);
    check_launch();
}

// ---------------------------------------------------------------------------------------------
// driver integrals over a prefix of cells (ref: src/examples/homogenized_coefficients.jl:592-667)
//   mode 0: sum_cells |J| sum_i v_i (b_i + (M v)_i), b = dphi . P = the right-hand side of outer step 0   integrate_first_term
//   mode 1: sum_cells |J| sum_i (v_i + w_i) (M v)_i                                                        integrate_terms
// M = reference-element mass matrix of the level.  One pass of the FUSED operator apply in "reductions only" form
// (no output vector): mass term only and unscaled, unit multiplicities, no constraint; the per-cell sum it leaves
// in blockpart is sum_i x_i (src_i + (M x)_i) resp. sum_i (x_i + src_i) (M x)_i, scaled by |J_c| in the reduction.
// Works for every level the apply works for (the slab kernel included) and moves 16 B per DOF.
// ---------------------------------------------------------------------------------------------
void launch_integrate(const Launch &L, const LevelDev &lv, const MeshDev &mesh, int mode, int64_t nsub, const double *v,
                      const double *second, int slot)
{
    ApplyArgs a{};
    a.alpha = 1.0;
    a.lambda = 1.0;
    a.x = v;
    a.src = second;
    a.flags = 2 | 16 | 32 | (mode == 1 ? 8 : 0);
    a.ncells_prefix = nsub;
    launch_apply_fused_kernel(L, lv, mesh, a);
    hipLaunchKernelGGL(k_reduce_weighted, dim3(256), dim3(256), 0, L.stream, mesh.blockpart, mesh.coef, lv.nterm - 1, nsub,
                       L.partials);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, L.stream, L.partials, 256, L.scal, slot);
    check_launch();
}

// ---------------------------------------------------------------------------------------------
// right-hand side F(v) = -int a xi . grad v  (ref: src/examples/homogenized_coefficients.jl:449-474)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_rhs_dphi(LevelDev lv, int64_t ncells, const double *__restrict__ pvec, double *b)
{
    const int64_t total = ncells * lv.nf;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i / lv.nf;
        const int t = (int)(i - c * lv.nf);
        const double *d = lv.dphi + 3 * t;
        const double *p = pvec + 3 * c;
        double v = d[0] * p[0];
        v += d[1] * p[1];
        if (lv.dim == 3) v += d[2] * p[2];
        b[c * lv.ld + t] = v;
    }
}

void launch_rhs_dphi(const Launch &L, const LevelDev &lv, int64_t ncells, const double *pvec, double *b)
{
    hipLaunchKernelGGL(k_rhs_dphi, dim3(strided_blocks(L, ncells * lv.nf, 4)), dim3(256), 0, L.stream, lv, ncells, pvec,
                       b);
    check_launch();
}

// ---------------------------------------------------------------------------------------------
// multi-GPU: pack one value per cut DOF (first local copy) / write the summed value to all copies -- faces, edges and
// nodes in ONE launch per direction (29 exchanges per V-cycle: three launches each were 1 ms of launch-bound kernels)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cut_pack(LevelDev lv, CutPackArgs c, double *buf, double *x, int unpack)
{
    const int64_t t0 = c.n[0] * lv.nfi, t1 = t0 + c.n[1] * lv.nei, total = t1 + c.n[2];
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int kind = idx < t0 ? 0 : idx < t1 ? 1 : 2;
        const int per = kind == 0 ? lv.nfi : kind == 1 ? lv.nei : 1;
        const int off = kind == 0 ? lv.off_face : kind == 1 ? lv.off_edge : 0;
        const int64_t r = idx - (kind == 0 ? 0 : kind == 1 ? t0 : t1);
        const int64_t e = r / per;
        const int k = (int)(r - e * per);
        const int32_t v = c.cell_lid[kind][e];
        double *a = x + (int64_t)(v >> 3) * lv.ld + off + (v & 7) * per + k;
        double *b = buf + c.pos[kind][e] + k;
        if (unpack)
            *a = *b;
        else if (c.first[kind][e])
            *b = *a;
    }
}

void launch_cut_pack(const Launch &L, const LevelDev &lv, const CutPackArgs &c, double *buf, double *x, int unpack)
{
    const int64_t total = c.n[0] * lv.nfi + c.n[1] * lv.nei + c.n[2];
    if (total == 0) return;
    hipLaunchKernelGGL(k_cut_pack, dim3(strided_blocks(L, total, 1)), dim3(256), 0, L.stream, lv, c, buf, x, unpack);
    check_launch();
}

// plan: nseg, then per segment (offset, size, members, offset of its member table); member table: staging offset of every
// member in ascending rank order, -1 for this rank
__global__ void __launch_bounds__(256) k_seg_sum(const int64_t *__restrict__ plan, int64_t total, double *buf,
                                                 const double *__restrict__ stage)
{
    const int nseg = (int)plan[0];
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int sgi = 0;
        while (sgi + 1 < nseg && idx >= plan[1 + 4 * (sgi + 1)]) ++sgi;        // (a few dozen segments at most)
        const int64_t off = plan[1 + 4 * sgi], d = idx - off;
        const int nm = (int)plan[3 + 4 * sgi];
        const int64_t *mt = plan + plan[4 + 4 * sgi];
        double acc = 0.0;
        for (int m = 0; m < nm; ++m) {
            const int64_t so = mt[m];
            const double v = so < 0 ? buf[idx] : stage[so + d];
            acc = m == 0 ? v : acc + v;
        }
        buf[idx] = acc;
    }
}

void launch_seg_sum(const Launch &L, const int64_t *plan, int64_t ndoubles, double *buf, const double *stage)
{
    if (ndoubles == 0) return;
    hipLaunchKernelGGL(k_seg_sum, dim3(strided_blocks(L, ndoubles, 1)), dim3(256), 0, L.stream, plan, ndoubles, buf, stage);
    check_launch();
}

}  // namespace hmg
