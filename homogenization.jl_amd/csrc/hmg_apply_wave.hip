// One wave per coarse cell: the operator apply of level 5 (3D, m = 16, 969 nodes per cell) for gfx950.
//
// Why a kernel of its own (round 4).  The 256-thread k_apply<3,256,4,*,4> carries ONE 7.75 KB column per workgroup; six
// workgroups are resident per CU, so 46 KB per stream are in flight per CU -- a third of what level 6 has -- and a
// workgroup's life is a chain of round trips (coefficients, class rows, column in two batches, addressing words, barrier,
// evaluation, two more barriers for the reductions) that is as long for 969 nodes as for 6545: 2.3-2.6 TB/s on the light
// passes.  Here a cell belongs to a single wave:
//   * 16 slots per lane and stream are requested back to back (16 waves per CU: 131 KB per stream in flight per CU);
//   * no barrier anywhere -- LDS operations of one wave execute in order, a lane may read what another lane of the same
//     wave wrote before;
//   * the waves are persistent (16 per CU walk the cells) and every table word a lane needs -- LDS positions of its 16
//     slots, addressing words of its 8 face runs, 2 edge/corner runs and 3 interior blocks -- is cell-independent, so it is
//     loaded ONCE per wave and stays in registers: no table traffic per cell at all (the 256-thread kernel re-reads ~5 KB
//     of L2-resident words and 14 KB of class-table terms per 15.5 KB column);
//   * the class weight rows do not depend on the cell but on its CLASS (cells with bitwise equal coefficient rows: at most
//     8 sigma triples x 6 orientations = 48 on a checkerboard): hmg_grid_set_operator forms W[class][sign][entity class][15]
//     once (k_weight_cache: the same seven products in the same order as the in-kernel form, so every weight -- and with it
//     every output value -- is bit-identical), and a wave fetches the rows it needs with SCALAR loads (constant address
//     space -> s_load_dwordx16) straight into SGPRs: no VALU work, no readlanes, no LDS table for faces and interior.
// Same arithmetic per node, in the same order, as k_apply (class_items / interior_block_core / stencil_eval_c): outputs are
// bit-identical (except on the three edges of the slanted face, whose absent tap (i+1,j,k) is multiplied by its zero weight
// here and left out at compile time there -- the backend then fuses the first two products the other way round: a last-bit
// difference, tests/test_gpu_wave.py); the per-cell partial sums of the fused CG pass (p.Ap, r.r) are formed over a different distribution of
// the nodes over the lanes and agree to rounding.
//
// Reference behaviour reproduced: src/apply_local_operators.jl:85-133 (+ :18-27 residual, constraint mask
// src/implicit_fine_grid.jl:94-139, the CG extras of src/multigrid.jl:54-68, restrict_to! / interpolate_and_sum_to!
// src/interpolation.jl:52-74 in the folded forms).
#include "hmg_device.hpp"
#include "hmg_stencil.hpp"

#include <algorithm>
#include <stdexcept>
#include <type_traits>

namespace hmg {

namespace {

// geometry this kernel is compiled for (checked by apply_wave_ok against the level's tables)
constexpr int WM = 16, WNF = 969, WNFI = 105, WNEI = 15, WNCORNER = 4, WNBLK = 152, WNFC = 165, WR = 4;
constexpr int WOFF_EDGE = 4, WOFF_FACE = 94, WNEC = 94;
constexpr int WNQ = 16;            // slots per lane: 15 full rounds of 64 + 9
constexpr int WVZ = 168;           // doubles in front of the lattice image: class rows 5..14 (edges, corners) x 16 + 8 spare (lanes without a slot write to the last one)

// a zero the backend cannot see through: OR-ed into a loop-invariant table word it makes everything DECODED from the word
// belong to the current cell -- otherwise the backend hoists dozens of decoded addresses out of the cell loop and spills them
// (lessons of the round-2 pipelined kernel) -- while the word itself stays an ordinary loop-invariant value
__device__ __forceinline__ uint32_t opaque_zero()
{
    uint32_t z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

// L | len << 10 | A << 15 | B << 23 | valid << 31 (host: build_wave_tables)
__device__ __forceinline__ void wdecode(uint32_t w, int &L, int &len, int &A, int &B)
{
    L = (int)(w & 1023u);
    len = (int)((w >> 10) & 31u);
    A = (int)((w >> 15) & 255u);
    B = (int)((w >> 23) & 255u);
}

// one node of an entity class whose weight row sits in SGPRs; M = the taps that exist (order and roundings of class_items)
template <uint32_t M>
__device__ __forceinline__ double eval_row(const double (&w)[15], const double *xs, uint32_t word, double &ctr)
{
    int L, len, A, B;
    wdecode(word, L, len, A, B);
    const double *p = xs + L;
    ctr = lds_ld(p);
    double acc = w[0] * ctr;
    if ((M >> 1) & 1u) acc += w[1] * lds_ld(p + 1);
    if ((M >> 2) & 1u) acc += w[2] * lds_ld(p - 1);
    if ((M >> 3) & 1u) acc += w[3] * lds_ld(p + len - 1);
    if ((M >> 4) & 1u) acc += w[4] * lds_ld(p - len);
    if ((M >> 5) & 1u) acc += w[5] * lds_ld(p + len);
    if ((M >> 6) & 1u) acc += w[6] * lds_ld(p - len - 1);
    const double *pu = p + A, *pd = p - B;
    if ((M >> 7) & 1u) acc += w[7] * lds_ld(pu - len);
    if ((M >> 8) & 1u) acc += w[8] * lds_ld(pd + len + 1);
    if ((M >> 9) & 1u) acc += w[9] * lds_ld(pu - 1);
    if ((M >> 10) & 1u) acc += w[10] * lds_ld(pd + 1);
    if ((M >> 11) & 1u) acc += w[11] * lds_ld(pu);
    if ((M >> 12) & 1u) acc += w[12] * lds_ld(pd);
    if ((M >> 13) & 1u) acc += w[13] * lds_ld(pu + 1 - len);
    if ((M >> 14) & 1u) acc += w[14] * lds_ld(pd + len);
    return acc;
}

// the two runs of 64 slots of face F (105 nodes): weight row from the cache by scalar loads
template <int F, bool FUSED, bool PAP, bool SRC>
__device__ __forceinline__ void wave_face(const double __attribute__((address_space(4))) *wrow, const double *xs, uint32_t wd0,
                                          uint32_t wd1, bool dirichlet, double mult, double sv0, double sv1, double *oc, double &pap,
                                          int lane, double &k0, double &k1)
{
    double w[15];
#pragma unroll
    for (int d = 0; d < 15; ++d) w[d] = (face_tap_mask(F) >> d) & 1u ? wrow[d] : 0.0;
    const int t0 = WOFF_FACE + F * WNFI + lane;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t wd = r ? wd1 : wd0;
        const int t = t0 + 64 * r;
        const bool valid = r == 0 || lane < WNFI - 64;
        double o = 0.0;
        if (!dirichlet) {
            const double sv = SRC ? (r ? sv1 : sv0) : 0.0;
            double ctr;
            const double acc = eval_row<face_tap_mask(F)>(w, xs, wd, ctr);
            o = sv + acc;
            if (PAP && valid) pap += mult * (ctr * o);
        }
        if (valid && (!FUSED || oc)) oc[t] = o;
        (r ? k1 : k0) = o;
    }
}

}  // namespace

// FUSED: the CG extras of k_apply (load phase: x-update of the previous step, p-update, two pending x-updates; epilogue:
// the cell's share of p.Ap and r.r).  CG: the coarse-grid correction x += P xcoarse folded into the load phase, the coarse
// column staged at the even nodes of the image.  RS: the results are restricted to the coarser level in the epilogue
// (summation order of k_restrict = the reference's).  SRC: out = src + alpha A x.
// (16 waves per CU fit the LDS and leave 128 VGPRs; the two instantiations that hold a second set of values across the cell --
//  the parents' words, or every result for the epilogue -- take 168 VGPRs and run 12 waves per CU instead of spilling)
template <bool FUSED, bool CG, bool RS, bool SRC>
__global__ void __launch_bounds__(64, (CG || RS || (SRC && FUSED)) ? 3 : 4)
k_apply_wave(LevelDev lv, const uint16_t *__restrict__ dmask, const int32_t *__restrict__ cell_class, ApplyArgs a)
{
    extern __shared__ double smem[];
    double *xs = smem + WVZ;
    char *sb = reinterpret_cast<char *>(smem);
    const int lane = threadIdx.x;
    const int64_t ld = lv.ld;

    // ---- once per wave: the lane's table words (cell-independent) and the zero guard behind the image
    uint32_t lpk[8], tw[WAVE_TAB_ROWS];
#pragma unroll
    for (int i = 0; i < 8; ++i) lpk[i] = lv.wave_lpos[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < WAVE_TAB_ROWS; ++i) tw[i] = lv.wave_tab[i * 64 + lane];
    uint32_t clo[3] = {0u, 0u, 0u};
    if constexpr (CG) {
#pragma unroll
        for (int q = 0; q < 3; ++q) clo[q] = lv.wave_cl[q * 64 + lane];
    }
    for (int q = lane; q < lv.lds_g1; q += 64) xs[WNF + q] = 0.0;
    if (lane < 8) smem[160 + lane] = 0.0;
    const bool neg = a.alpha < 0.0;
    const int t15 = lane + 960 < WNF ? lane + 960 : WNF - 1;   // (the 16th round: 9 lanes have a slot, the others reload the last one)
    const bool v15 = lane + 960 < WNF;
    double beta = 0.0, ax = 0.0, c2 = 0.0;
    if constexpr (FUSED) {
        beta = a.x2 ? a.scal[a.s_num] / a.scal[a.s_den] : 0.0;
        ax = a.xacc || a.x3 ? a.scal[a.a_num] / a.scal[a.a_den] : 0.0;
        c2 = a.x3 ? a.scal[a.c_num] / a.scal[a.c_den] : 0.0;
    }
    // (one opaque zero per USE: a shared one lets the backend form every `word | zero` at the top of the cell and keep
    //  them all -- the table twice)
    auto lds_off = [&](int q) -> uint32_t {
        const uint32_t w = lpk[q >> 1] | opaque_zero();
        return (q & 1) ? w >> 16 : w & 0xffffu;
    };
    auto put = [&](int q, double v) { *reinterpret_cast<double *>(sb + lds_off(q)) = v; };
    auto TW = [&](int i) -> uint32_t { return tw[i] | opaque_zero(); };

    for (int64_t blk = blockIdx.x; blk < a.nwork; blk += gridDim.x) {
        const int64_t cell = a.cell_list ? (int64_t)HMG_KP(int32_t, a.cell_list)[blk] : blk;
        const int wsel = 2 * HMG_KP(int32_t, cell_class)[cell] + (neg ? 1 : 0);
        const double __attribute__((address_space(4))) *wc = HMG_KP(double, lv.wcache) + (size_t)wsel * WAVE_WSTRIDE;
        uint32_t dm = 0u;
        if (a.flags & 1) {
            const uint32_t w2 = HMG_KP(uint32_t, dmask)[cell >> 1];
            dm = (cell & 1) ? w2 >> 16 : w2 & 0xffffu;
        }
        uint32_t mq[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
        if (FUSED && a.mult) {
#pragma unroll
            for (int q = 0; q < 4; ++q) mq[q] = HMG_KP(uint32_t, a.mult)[cell * 4 + q];
        }
        // the rows of the edge and corner classes (5..14) go to LDS: those nodes are evaluated with one class per LANE
        const double *wcg = lv.wcache + (size_t)wsel * WAVE_WSTRIDE + 5 * WAVE_ROW;
        const double we0 = wcg[lane], we1 = wcg[lane + 64], we2 = wcg[128 + (lane & 31)];

        const double *xc = a.x + cell * ld;
        double rr = 0.0, pap = 0.0;
        // SRC: the source values of the nodes this lane will EVALUATE are requested ahead of their use instead of one exposed
        // round trip per run in the middle of the evaluation (first version: 1.85 ms for the 33 B/DOF passes, 3.4 TB/s): those
        // of the 8 face and 2 edge/corner runs behind the last batch of column loads (they arrive in order behind the column),
        // those of the 3 interior blocks at the start of the evaluation, when the column's registers are free again -- the
        // face runs hide them
        const double *sc = SRC ? a.src + cell * ld : nullptr;
        double sf[8], sec[2], si[3][WR];
        auto issue_src = [&]() {
            if constexpr (SRC) {
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int t0 = WOFF_FACE + f * WNFI + lane;
                    sf[2 * f] = sc[t0];
                    sf[2 * f + 1] = sc[lane < WNFI - 64 ? t0 + 64 : t0];
                }
                sec[0] = sc[lane];
                sec[1] = sc[64 + lane];                          // (slots 94..127 exist: face nodes -- loaded, not used)
            }
        };
        auto issue_src_interior = [&]() {
            if constexpr (SRC) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int slot0 = (int)(r == 0 ? TW(13) & 0xffffu : r == 1 ? TW(13) >> 16 : TW(14) & 0xffffu);
                    int slot[WR];
                    block_slots<WR>(WM, TW(10 + r), slot0, slot);
                    const int nv = (int)(TW(10 + r) >> 28);
#pragma unroll
                    for (int q = 0; q < WR; ++q) si[r][q] = sc[q < nv ? slot[q] : slot0];
                }
            }
        };
        // ---- load phase: the column goes to the lattice image; FUSED: the side effects of the CG step ride along
        if constexpr (!FUSED) {
            double xv[WNQ];
#pragma unroll
            for (int q = 0; q < WNQ - 1; ++q) xv[q] = xc[lane + 64 * q];
            xv[WNQ - 1] = xc[t15];
            issue_src();
            smem[lane] = we0;
            smem[lane + 64] = we1;
            smem[128 + (lane & 31)] = we2;
#pragma unroll
            for (int q = 0; q < WNQ; ++q) put(q, xv[q]);
        } else {
            const double *x2c = a.x2 ? a.x2 + cell * ld : nullptr;
            double *xoc = a.xout ? a.xout + cell * ld : nullptr;
            double *xac = a.xacc ? a.xacc + cell * ld : nullptr;
            const double *x3c = a.x3 ? a.x3 + cell * ld : nullptr;
            const bool xzero = (a.flags & 128) != 0;
            if constexpr (CG) {
                // coarse-grid correction (interpolate_and_sum_to!, src/interpolation.jl:64-74): the coarse column goes to the
                // EVEN nodes of the image, every lane combines its slots' parents from there (identity rows x + c_a,
                // midpoints (x + 0.5 c_a) + 0.5 c_b: the reference's roundings), then the image takes the fine values
                const double *ccol = a.xcoarse + cell * a.ldc;
                double cv[3], xv[WNQ];
                uint32_t pw[WNQ];
#pragma unroll
                for (int q = 0; q < 3; ++q) cv[q] = ccol[lane + 64 * q < WNFC ? lane + 64 * q : WNFC - 1];
#pragma unroll
                for (int q = 0; q < WNQ - 1; ++q) xv[q] = xc[lane + 64 * q];
                xv[WNQ - 1] = xc[t15];
#pragma unroll
                for (int q = 0; q < WNQ; ++q) pw[q] = lv.wave_par[q * 64 + lane];
                issue_src();
                smem[lane] = we0;
                smem[lane + 64] = we1;
                smem[128 + (lane & 31)] = we2;
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<double *>(sb + (clo[q] | opaque_zero())) = cv[q];
#pragma unroll
                for (int q = 0; q < WNQ; ++q) {
                    // (both forms computed, one selected: no branch per slot)
                    const uint32_t pa = pw[q] & 0xffffu, pb = pw[q] >> 16;
                    const double ca = lds_ld(xs + pa), cb = lds_ld(xs + pb);
                    const double vi = xv[q] + ca;
                    double vm = xv[q];
                    vm += 0.5 * ca;
                    vm = vm + 0.5 * cb;
                    xv[q] = pa == pb ? vi : vm;
                }
#pragma unroll
                for (int q = 0; q < WNQ; ++q) {
                    const bool valid = q < WNQ - 1 || v15;
                    const double v = valid ? xv[q] : 0.0;
                    if (valid) xoc[q < WNQ - 1 ? lane + 64 * q : t15] = v;     // (xout is there: apply_wave_ok)
                    rr += v * v;
                    put(q, v);
                }
            } else if (!x2c && !xac && !x3c) {
                // CG step 0 with r itself as p: one stream, the whole column in one batch
                double xv[WNQ];
#pragma unroll
                for (int q = 0; q < WNQ - 1; ++q) xv[q] = xc[lane + 64 * q];
                xv[WNQ - 1] = xc[t15];
                issue_src();
                smem[lane] = we0;
                smem[lane + 64] = we1;
                smem[128 + (lane & 31)] = we2;
#pragma unroll
                for (int q = 0; q < WNQ; ++q) {
                    const bool valid = q < WNQ - 1 || v15;
                    const double v = valid ? xv[q] : 0.0;
                    if (valid && xoc) xoc[q < WNQ - 1 ? lane + 64 * q : t15] = v;
                    rr += v * v;
                    put(q, v);
                }
            } else {
                // general forms: all loads of a batch back to back, then its stores (a store to xout / xacc, which may alias
                // x2, never sits between two loads of a batch).  One straight-line variant per combination of streams -- a
                // test of an optional pointer between two loads would end the batch (the backend waits at every join).
                smem[lane] = we0;
                smem[lane + 64] = we1;
                smem[128 + (lane & 31)] = we2;
                auto general = [&](auto acc_c, auto x3_c, auto xz_c, auto hb_c) {
                    constexpr bool ACC = decltype(acc_c)::value, X3 = decltype(x3_c)::value, XZ = decltype(xz_c)::value;
                    constexpr int HB = decltype(hb_c)::value;
#pragma unroll
                    for (int q0 = 0; q0 < WNQ; q0 += HB) {
                        double xv[HB], x2v[HB], xav[HB];
#pragma unroll
                        for (int q = 0; q < HB; ++q) {
                            const int t = q0 + q < WNQ - 1 ? lane + 64 * (q0 + q) : t15;
                            xv[q] = XZ ? 0.0 : xc[t];
                            x2v[q] = x2c[t];
                            xav[q] = ACC ? xac[t] : X3 ? x3c[t] : 0.0;
                        }
                        if (q0 + HB >= WNQ) issue_src();
#pragma unroll
                        for (int q = 0; q < HB; ++q) {
                            const bool valid = q0 + q < WNQ - 1 || v15;
                            const int t = q0 + q < WNQ - 1 ? lane + 64 * (q0 + q) : t15;
                            double v = xv[q];
                            if (ACC && valid) xac[t] = axpy1(ax, x2v[q], xav[q]);
                            if (X3) {
                                const double t1 = axpy1(ax, x2v[q], v);
                                const double p2 = axpy1(beta, x2v[q], xav[q]);
                                v = axpy1(c2, p2, t1);
                            } else
                                v = axpy1(beta, x2v[q], v);
                            if (valid && ((ACC || X3) || xoc)) xoc[t] = v;   // (ACC / X3: xout is there, apply_wave_ok)
                            if (!valid) v = 0.0;
                            if (!RS) rr += v * v;
                            put(q0 + q, v);
                        }
                    }
                };
                using T_ = std::true_type;
                using F_ = std::false_type;
                if (x3c) {          // two pending x-updates folded into a residual (k_apply's x3 mode); x may be a zero nobody wrote
                    if (xzero)
                        general(F_{}, T_{}, T_{}, std::integral_constant<int, 8>{});
                    else
                        general(F_{}, T_{}, F_{}, std::integral_constant<int, 8>{});
                } else if (xac)     // full CG step: x += alpha p_old; p = r + beta p_old
                    general(T_{}, F_{}, F_{}, std::integral_constant<int, 8>{});
                else                // p = r + beta p_old only (dead / last step: two streams, the whole column in one batch)
                    general(F_{}, F_{}, F_{}, std::integral_constant<int, 16>{});
            }
        }

        // ---- evaluation.  Faces and interior: class rows in SGPRs; edges and corners: one class per lane, rows in LDS
        double *oc = a.out ? a.out + cell * ld : nullptr;
        double kf[8], kec[2], ki[3][WR];
        // (scheduling fences: without them the backend hoists these loads -- and the evaluation's address arithmetic -- into the
        //  load phase, whose batches already fill the register budget)
        __builtin_amdgcn_sched_barrier(0);
        issue_src_interior();
        __builtin_amdgcn_sched_barrier(0);
        {
            const double fm0 = (double)(mq[0] & 0xffu), fm1 = (double)((mq[0] >> 8) & 0xffu), fm2 = (double)((mq[0] >> 16) & 0xffu),
                         fm3 = (double)(mq[0] >> 24);
            wave_face<0, FUSED, FUSED && !RS, SRC>(wc + 1 * WAVE_ROW, xs, TW(0), TW(1), (dm >> 0) & 1u, fm0, sf[0], sf[1], oc, pap, lane, kf[0], kf[1]);
            wave_face<1, FUSED, FUSED && !RS, SRC>(wc + 2 * WAVE_ROW, xs, TW(2), TW(3), (dm >> 1) & 1u, fm1, sf[2], sf[3], oc, pap, lane, kf[2], kf[3]);
            wave_face<2, FUSED, FUSED && !RS, SRC>(wc + 3 * WAVE_ROW, xs, TW(4), TW(5), (dm >> 2) & 1u, fm2, sf[4], sf[5], oc, pap, lane, kf[4], kf[5]);
            wave_face<3, FUSED, FUSED && !RS, SRC>(wc + 4 * WAVE_ROW, xs, TW(6), TW(7), (dm >> 3) & 1u, fm3, sf[6], sf[7], oc, pap, lane, kf[6], kf[7]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {          // corners (slots 0..3) and edges (4..93): the lane's class row is read from LDS tap by tap
            const int t = r * 64 + lane;
            const bool valid = t < WNEC;
            const int cls = (int)((TW(14) >> (16 + 8 * r)) & 0xffu);
            const double sv = SRC ? sec[r] : 0.0;
            int L, len, A, B;
            wdecode(TW(8 + r), L, len, A, B);
            double ctr;
            double o = stencil_eval_c<3>(smem + (cls - 5) * WAVE_ROW, xs, L, len, A, B, ctr);
            o = sv + o;
            if ((dm >> (cls - 1)) & 1u) o = 0.0;
            if (valid && (!FUSED || oc)) oc[t] = o;
            if (FUSED && !RS && valid) {
                const int e = cls - 1;
                const uint32_t word = e < 4 ? mq[0] : e < 8 ? mq[1] : e < 12 ? mq[2] : mq[3];
                const uint32_t mu = (word >> (8 * (e & 3))) & 0xffu;
                pap += (double)mu * (ctr * o);
            }
            kec[r] = o;
        }
        {
            double w0[15];
#pragma unroll
            for (int d = 0; d < 15; ++d) w0[d] = wc[d];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const uint32_t word = TW(10 + r);
                const int slot0 = (int)(r == 0 ? TW(13) & 0xffffu : r == 1 ? TW(13) >> 16 : TW(14) & 0xffffu);
                if (r < 2 || lane < WNBLK - 128) {
                    // (the sums start from the source values -- 0 without a source, and for the nodes a block does not have)
                    const int nv = (int)(word >> 28);
#pragma unroll
                    for (int q = 0; q < WR; ++q) ki[r][q] = SRC && q < nv ? si[r][q] : 0.0;
                    interior_block_core<WR, FUSED, false, RS>(w0, xs, WM, WNF >> 1, word, slot0, oc, pap, ki[r]);
                }
            }
        }

        if constexpr (RS) {
            // Restriction in the epilogue (restrict_to!, src/interpolation.jl:52-62, of the cell-local residual just formed --
            // src/multigrid.jl:104-105): every read of the image is done (the results depend on them), so the results replace
            // the image and the coarse right-hand side is summed from there in the order of k_restrict -- identity row first,
            // then the midpoints in ascending fine hierarchical id, the reference's CSC order: the same bits.
#pragma unroll
            for (int f = 0; f < 8; ++f)
                if (!(f & 1) || lane < WNFI - 64) xs[TW(f) & 1023u] = kf[f];
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (r * 64 + lane < WNEC) xs[TW(8 + r) & 1023u] = kec[r];
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (r < 2 || lane < WNBLK - 128) {
                    int pos[WR];
                    block_positions<WR>(WM, TW(10 + r), pos);
                    const int nv = (int)(TW(10 + r) >> 28);
#pragma unroll
                    for (int q = 0; q < WR; ++q)
                        if (q < nv) xs[pos[q]] = ki[r][q];
                }
            double *rc = a.rcoarse + cell * a.ldrc;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int c = lane + 64 * q;
                if (c < WNFC) {
                    const uint4 ta = reinterpret_cast<const uint4 *>(lv.wave_rs)[2 * c];
                    const uint4 tb = reinterpret_cast<const uint4 *>(lv.wave_rs)[2 * c + 1];
                    const uint32_t e[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
                    const int n = (int)(e[7] >> 16);
                    double tmp = 0.0;
                    tmp += 1.0 * lds_ld(xs + (e[0] & 0xffffu));
#pragma unroll
                    for (int u = 1; u < 15; ++u) {       // (entries beyond n address position 0: read, not added)
                        const double t2 = tmp + 0.5 * lds_ld(xs + ((u & 1) ? e[u >> 1] >> 16 : e[u >> 1] & 0xffffu));
                        tmp = u < n ? t2 : tmp;
                    }
                    rc[c] = tmp;
                }
            }
        }
        if (FUSED && !RS) {
            const double s_pap = wave_sum63(pap), s_rr = wave_sum63(rr);
            if (lane == 63) {
                a.blockpart[2 * cell] = s_pap;
                a.blockpart[2 * cell + 1] = s_rr;
            }
        }
    }
}

// W[class][sign][entity class][16]: the weights k_apply forms per cell (cell_scales + the seven-term sums, same order --
// bit-identical), once per distinct coefficient row
__global__ void __launch_bounds__(256)
k_weight_cache(LevelDev lv, const double *__restrict__ coef_rep, double lambda, double *__restrict__ wcache)
{
    const int cls = blockIdx.x >> 1;
    const double alpha = (blockIdx.x & 1) ? -1.0 : 1.0;
    double s[7];
    cell_scales<3>(coef_rep + (size_t)cls * 8, alpha, lambda, s, 0);
    double *out = wcache + (size_t)blockIdx.x * WAVE_WSTRIDE;
    for (int idx = threadIdx.x; idx < 15 * WAVE_ROW; idx += 256) {
        const int c = idx / WAVE_ROW, d = idx % WAVE_ROW;
        double w = 0.0;
        if (c < lv.ncls && d < 15) {
            const double *ct = lv.ctab + ((size_t)c * 15 + d) * 7;
#pragma unroll
            for (int t = 0; t < 7; ++t) w += ct[t] * s[t];
        }
        out[idx] = w;
    }
}

void launch_weight_cache(const Launch &L, const LevelDev &lv, const double *coef_rep, int nclasses, double lambda, double *wcache)
{
    if (nclasses <= 0) return;
    if (lv.dim != 3 || lv.ncls != 15 || lv.ndir != 15 || lv.nterm != 7) throw std::runtime_error("weight cache: not a 3D level");
    hipLaunchKernelGGL(k_weight_cache, dim3((unsigned)(2 * nclasses)), dim3(256), 0, L.stream, lv, coef_rep, lambda, wcache);
    check_launch();
}

bool apply_wave_ok(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a, bool fused)
{
    if (!L.apply_wave || L.apply_threads != 0 || L.apply_unblocked) return false;
    if (!lv.wave_tab || !lv.wave_lpos || !lv.wcache || !mesh.cell_class) return false;
    if (lv.dim != 3 || lv.m != WM || lv.nf != WNF || lv.nfi != WNFI || lv.nei != WNEI || lv.ncorner != WNCORNER || lv.nedge != 6 ||
        lv.nface != 4 || lv.nblk != WNBLK || lv.blk_R != WR || lv.off_edge != WOFF_EDGE || lv.off_face != WOFF_FACE ||
        lv.lds_g0 != 0 || lv.lds_g1 < WM + 8 || lv.lds_g1 > 64 || lv.ncls != 15)
        return false;
    if (a.flags & (2 | 8 | 16)) return false;                       // mass-only / driver-integral forms: weights not cached
    if (!(a.alpha == 1.0 || a.alpha == -1.0) || a.lambda != mesh.wc_lambda) return false;
    if (!fused) return !a.xcoarse && !a.rcoarse;
    if (a.xcoarse && a.rcoarse) return false;
    if ((a.x3 || a.xacc) && !(a.x2 && a.xout)) return false;          // (those forms store through xout unconditionally)
    if (a.xcoarse) return a.src && a.xout && lv.wave_par && lv.wave_cl && lv.nf_coarse == WNFC && !a.x2 && !a.xacc && !a.x3;
    if (a.rcoarse) return a.src && lv.wave_rs && lv.nf_coarse == WNFC;
    return true;
}

void launch_apply_wave(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a0, bool fused)
{
    ApplyArgs a = a0;
    const int64_t nblocks = a.cell_list ? a.ncell_list : a.ncells_prefix ? a.ncells_prefix : mesh.ncells;
    if (nblocks == 0) return;
    if (!a.x) throw std::runtime_error("operator apply: null input vector");
    if (!fused && !a.out) throw std::runtime_error("operator apply: a plain launch needs an output vector");
    if (fused && (!a.blockpart || !a.scal)) throw std::runtime_error("operator apply: fused launch without its reduction scratch");
    if ((a.flags & 1) && !mesh.dmask) throw std::runtime_error("operator apply: constraint requested without a Dirichlet mask");
    if (a.xcoarse && !a.xout) throw std::runtime_error("operator apply: folded prolongation without xout");
    if (a.rcoarse && a.ldrc <= 0) throw std::runtime_error("operator apply: epilogue restriction without the coarse column stride");
    if ((a.flags & 128) && !(fused && a.x3 && a.x2 && a.xout && !a.xcoarse))
        throw std::runtime_error("operator apply: the zero-input form exists for the residual with two pending x-updates only");
    a.nwork = nblocks;
    if (L.cell_order && !a.cell_list && !a.ncells_prefix && mesh.cell_perm) {   // XCD x walks the x-th eighth of the cells
        a.cell_list = mesh.cell_perm;
        a.ncell_list = nblocks;
    }
    int64_t grid = std::min<int64_t>(nblocks, L.wave_grid > 0 ? L.wave_grid : 16 * (int64_t)L.num_cu);
    if (grid >= 8) grid -= grid % 8;               // (wave b stays on XCD b % 8 for every cell it walks)
    const size_t lds = sizeof(double) * (size_t)(WVZ + WNF + lv.lds_g1);
    const dim3 g((unsigned)grid), b(64);
#define HMG_WAVE_LAUNCH(F, C, R, S) \
    hipLaunchKernelGGL((k_apply_wave<F, C, R, S>), g, b, lds, L.stream, lv, mesh.dmask, mesh.cell_class, a)
    if (!fused) {
        if (a.src)
            HMG_WAVE_LAUNCH(false, false, false, true);
        else
            HMG_WAVE_LAUNCH(false, false, false, false);
    } else if (a.xcoarse)
        HMG_WAVE_LAUNCH(true, true, false, true);
    else if (a.rcoarse)
        HMG_WAVE_LAUNCH(true, false, true, true);
    else if (a.src)
        HMG_WAVE_LAUNCH(true, false, false, true);
    else
        HMG_WAVE_LAUNCH(true, false, false, false);
#undef HMG_WAVE_LAUNCH
    check_launch();
    if (L.n_wave_launches) *L.n_wave_launches += 1;
}

}  // namespace hmg
