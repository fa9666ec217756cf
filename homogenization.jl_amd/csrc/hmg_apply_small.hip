// The operator apply of the SMALL 3D levels (2, 3, 4: 10, 35, 165 nodes per cell) for gfx950: one persistent wave per cell,
// software-pipelined over the cells it walks.
//
// Why (round 4).  Levels 2-4 hold 0.6 % of the DOFs and took 6 % of a V-cycle: k_apply<3,64,*> gives a cell to one wave, and
// a cell's life there is a chain of dependent round trips -- coefficients, 1575 class-table terms (12.6 KB for a 1.3 KB
// column), the column, the barrier, the evaluation, two block reductions -- about 8 us for a few hundred instructions, 24
// cells one after the other per wave.  Here
//   * the cell's class table comes from the class-weight cache (hmg_apply_wave.hip: one row set per distinct coefficient
//     row, formed at hmg_grid_set_operator): 240 values, four loads per lane;
//   * every node is evaluated by the lane that loaded its slot (slot t = lane + 64 q, class row read from LDS tap by tap), so
//     source values and results share the load mapping and ALL of a cell's global loads -- class table, up to three column
//     streams, source, coarse column -- are requested together;
//   * those loads are issued for the NEXT cell before the current one is evaluated (its values have gone to LDS by then: one
//     set of registers), so the memory latency of cell n + 1 hides behind the evaluation of cell n; the class and mask words
//     (scalar loads) run one more cell ahead;
//   * no barrier, no LDS reduction: one wave, ordered LDS operations, wave-wide sums by cross-lane moves.
// Same arithmetic per node as k_apply's surface path (stencil_eval_c); interior nodes take their row from LDS instead of SGPRs.
//
// ref: src/apply_local_operators.jl:85-133 (+ :18-27, constraint src/implicit_fine_grid.jl:94-139, CG extras src/multigrid.jl:54-68,
// interpolate_and_sum_to! src/interpolation.jl:64-74 in the folded form)
#include "hmg_device.hpp"
#include "hmg_stencil.hpp"

#include <algorithm>
#include <stdexcept>

namespace hmg {

namespace {
constexpr int SW = 15 * WAVE_ROW;      // doubles of LDS in front of the image: the cell's class table, rows of 16

__device__ __forceinline__ uint32_t small_opaque_zero()
{
    uint32_t z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
}  // namespace

template <int SPT, bool FUSED>
__global__ void __launch_bounds__(64, SPT == 1 ? 5 : 4)   // (107 / 88 VGPRs without scratch: 16 / 20 waves per CU -- the prefetch hides the latency, not the occupancy)
k_apply_small(LevelDev lv, const uint16_t *__restrict__ dmask, const int32_t *__restrict__ cell_class, ApplyArgs a)
{
    extern __shared__ double smem[];
    double *W = smem;
    double *xs = smem + SW + lv.lds_g0;
    const int lane = threadIdx.x, nf = lv.nf, m = lv.m;
    double *cs = xs + nf + lv.lds_g1;            // coarse column of the folded prolongation (nf_coarse <= 64 doubles)
    const int64_t ld = lv.ld;

    // ---- once per wave: table words of the lane's slots, zero guards, the CG ratios
    uint32_t pw[SPT], par[SPT];
    int lp[SPT];
    bool has[SPT];
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int t = lane + 64 * q;
        has[q] = t < nf;
        const int tc = has[q] ? t : nf - 1;
        pw[q] = lv.pos32[tc];
        lp[q] = (int)lv.lpos[tc];
        par[q] = FUSED && a.xcoarse ? lv.par32[tc] : 0u;
    }
    for (int q = lane; q < lv.lds_g0; q += 64) smem[SW + q] = 0.0;
    for (int q = lane; q < lv.lds_g1; q += 64) xs[nf + q] = 0.0;
    const bool neg = a.alpha < 0.0;
    double beta = 0.0, ax = 0.0, c2 = 0.0;
    if constexpr (FUSED) {
        beta = a.x2 ? a.scal[a.s_num] / a.scal[a.s_den] : 0.0;
        ax = a.xacc || a.x3 ? a.scal[a.a_num] / a.scal[a.a_den] : 0.0;
        c2 = a.x3 ? a.scal[a.c_num] / a.scal[a.c_den] : 0.0;
    }
    const bool xzero = FUSED && (a.flags & 128);
    const int64_t G = gridDim.x;
    auto cell_of = [&](int64_t blk) -> int64_t { return a.cell_list ? (int64_t)HMG_KP(int32_t, a.cell_list)[blk] : blk; };

    // ---- the loads of one cell (everything it reads from global memory), into one set of registers
    double pwv[4], pxv[SPT], px2[SPT], pxa[SPT], psv[SPT], pcv = 0.0;
    auto issue = [&](int64_t cell, int wsel) {
        const double *wc = lv.wcache + (size_t)wsel * WAVE_WSTRIDE;
#pragma unroll
        for (int q = 0; q < 4; ++q) pwv[q] = wc[lane + 64 * q < SW ? lane + 64 * q : SW - 1];
        const double *xc = a.x + cell * ld;
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int t = has[q] ? lane + 64 * q : nf - 1;
            pxv[q] = xzero ? 0.0 : xc[t];
            if (FUSED && a.x2) px2[q] = a.x2[cell * ld + t];
            if (FUSED && a.xacc)
                pxa[q] = a.xacc[cell * ld + t];
            else if (FUSED && a.x3)
                pxa[q] = a.x3[cell * ld + t];
            if (a.src) psv[q] = a.src[cell * ld + t];
        }
        if (FUSED && a.xcoarse) pcv = a.xcoarse[cell * a.ldc + (lane < lv.nf_coarse ? lane : lv.nf_coarse - 1)];
    };
    // the wave-uniform words of a cell: its class in the cache, Dirichlet mask, multiplicities (scalar loads)
    struct Uni {
        int wsel;
        uint32_t dm, mq[4];
    };
    auto uniform_of = [&](int64_t cell) {
        Uni u;
        u.wsel = 2 * HMG_KP(int32_t, cell_class)[cell] + (neg ? 1 : 0);
        u.dm = 0u;
        if (a.flags & 1) {
            const uint32_t w2 = HMG_KP(uint32_t, dmask)[cell >> 1];
            u.dm = (cell & 1) ? w2 >> 16 : w2 & 0xffffu;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) u.mq[q] = FUSED && a.mult ? HMG_KP(uint32_t, a.mult)[cell * 4 + q] : 0x01010101u;
        return u;
    };

    int64_t blk = blockIdx.x;
    if (blk >= a.nwork) return;
    int64_t cell = cell_of(blk);
    Uni cur = uniform_of(cell);
    issue(cell, cur.wsel);
    int64_t cell_n = blk + G < a.nwork ? cell_of(blk + G) : cell;
    Uni nxt = uniform_of(cell_n);

    for (;;) {
        // ---- the cell's values leave the load registers: class table and column to LDS, side effects of the CG step stored
        double rr = 0.0, pap = 0.0, sv[SPT];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (lane + 64 * q < SW) W[lane + 64 * q] = pwv[q];
        double v[SPT];
        if constexpr (FUSED) {
            if (a.xcoarse && lane < lv.nf_coarse) cs[lane] = pcv;
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int t = lane + 64 * q;
                double val = pxv[q];
                if (a.xcoarse) {        // interpolate_and_sum_to!: identity rows += 1.0 c[a], midpoints += 0.5 c[a], += 0.5 c[b]
                    const uint32_t pa = par[q] & 0xffffu, pb = par[q] >> 16;
                    const double ca = lds_ld(cs + pa), cb = lds_ld(cs + pb);
                    const double vi = val + ca;
                    double vm = val;
                    vm += 0.5 * ca;
                    vm = vm + 0.5 * cb;
                    val = pa == pb ? vi : vm;
                }
                if (a.xacc && has[q]) a.xacc[cell * ld + t] = axpy1(ax, px2[q], pxa[q]);
                if (a.x3) {
                    const double t1 = axpy1(ax, px2[q], val);
                    const double p2 = axpy1(beta, px2[q], pxa[q]);
                    val = axpy1(c2, p2, t1);
                } else if (a.x2)
                    val = axpy1(beta, px2[q], val);
                if (a.xout && has[q]) a.xout[cell * ld + t] = val;
                if (!has[q]) val = 0.0;
                rr += val * val;
                v[q] = val;
            }
        } else {
#pragma unroll
            for (int q = 0; q < SPT; ++q) v[q] = pxv[q];
        }
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            if (has[q]) xs[lp[q]] = v[q];
            sv[q] = a.src ? psv[q] : 0.0;
        }
        const Uni me = cur;
        const int64_t mycell = cell;
        // ---- the next cell's loads go out now; its uniform words were fetched a cell ago, the ones after it are fetched here
        const bool more = blk + G < a.nwork;
        if (more) {
            issue(cell_n, nxt.wsel);
            cell = cell_n;
            cur = nxt;
            cell_n = blk + 2 * G < a.nwork ? cell_of(blk + 2 * G) : cell_n;
            nxt = uniform_of(cell_n);
        }
        // ---- evaluation: every lane its own slots, class row from the LDS table
        double *oc = a.out ? a.out + mycell * ld : nullptr;
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const uint32_t word = pw[q] | small_opaque_zero();       // (keeps the decoded addresses inside the cell loop)
            int L, len, A, B, cls;
            decode32<3>(word, m, L, len, A, B, cls);
            double ctr;
            double o = stencil_eval_c<3>(W + cls * WAVE_ROW, xs, L, len, A, B, ctr);
            o = sv[q] + o;
            if (cls > 0 && ((me.dm >> (cls - 1)) & 1u)) o = 0.0;
            if (has[q] && (!FUSED || oc)) oc[lane + 64 * q] = o;
            if (FUSED && has[q]) {
                uint32_t mu = 1u;
                if (cls > 0) {
                    const int e = cls - 1;
                    const uint32_t wd = e < 4 ? me.mq[0] : e < 8 ? me.mq[1] : e < 12 ? me.mq[2] : me.mq[3];
                    mu = (wd >> (8 * (e & 3))) & 0xffu;
                }
                pap += (double)mu * (ctr * o);
            }
        }
        if constexpr (FUSED) {
            const double s_pap = wave_sum63(pap), s_rr = wave_sum63(rr);
            if (lane == 63) {
                a.blockpart[2 * mycell] = s_pap;
                a.blockpart[2 * mycell + 1] = s_rr;
            }
        }
        if (!more) break;
        blk += G;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Cells of at most 16 nodes (3D level 2: 10): FOUR cells per wave, one per row of 16 lanes.  With one cell per wave 10 of 64 lanes
// worked and level 2 cost as much per cell as level 3; a wave's instruction stream is the same for one cell or four.
//   * the lane's slot (its lattice position, class, tap offsets, restriction parents) is the same for every cell: decoded once;
//   * class, Dirichlet mask and multiplicity word of the lane's cell are per-lane loads, fetched TWO quads ahead; the lane's class
//     row (15 weights, straight from the class-weight cache into registers -- no LDS table) and its column values ONE quad ahead;
//   * the per-cell sums are the first four stages of wave_sum63 (row_shr 1, 2, 4, 8: the row's sum in its lane 15) -- the bits
//     k_apply_small's whole-wave sum gives for a cell that sits in row 0 with zeros behind it.
// Same arithmetic per node as k_apply_small (the tests compare the two bit for bit).
template <bool FUSED>
__global__ void __launch_bounds__(64, 4)
k_apply_pack(LevelDev lv, const uint16_t *__restrict__ dmask, const int32_t *__restrict__ cell_class, ApplyArgs a)
{
    extern __shared__ double smem[];
    const int lane = threadIdx.x, sub = lane >> 4, t = lane & 15, nf = lv.nf, m = lv.m;
    const bool has = t < nf;
    const int tc = has ? t : nf - 1;
    const int imgp = lv.lds_g0 + nf + lv.lds_g1;
    double *xs = smem + sub * imgp + lv.lds_g0;
    const int64_t ld = lv.ld;
    for (int q = lane; q < 4 * imgp; q += 64) smem[q] = 0.0;       // (guards stay zero; the images are rewritten per quad)

    int L, len, A, B, cls;
    decode32<3>(lv.pos32[tc], m, L, len, A, B, cls);
    const int lp = (int)lv.lpos[tc];
    const uint32_t par = FUSED && a.xcoarse ? lv.par32[tc] : 0u;
    const int pa = (int)(par & 0xffffu), pb = (int)(par >> 16);
    const int e = cls > 0 ? cls - 1 : 0;
    const bool neg = a.alpha < 0.0;
    double beta = 0.0, ax = 0.0, c2 = 0.0;
    if constexpr (FUSED) {
        beta = a.x2 ? a.scal[a.s_num] / a.scal[a.s_den] : 0.0;
        ax = a.xacc || a.x3 ? a.scal[a.a_num] / a.scal[a.a_den] : 0.0;
        c2 = a.x3 ? a.scal[a.c_num] / a.scal[a.c_den] : 0.0;
    }
    const bool xzero = FUSED && (a.flags & 128);
    const int64_t G = gridDim.x, nquad = (a.nwork + 3) >> 2;

    struct Meta {
        int cell, wsel;
        uint32_t dm, mw;
        bool valid;
    };
    auto meta_of = [&](int64_t Q) {
        Meta mt;
        int64_t w = 4 * Q + sub;
        mt.valid = w < a.nwork;
        if (!mt.valid) w = a.nwork - 1;
        mt.cell = a.cell_list ? a.cell_list[w] : (int)w;
        mt.wsel = 2 * cell_class[mt.cell] + (neg ? 1 : 0);
        mt.dm = (a.flags & 1) ? (uint32_t)dmask[mt.cell] : 0u;
        mt.mw = FUSED && a.mult && cls > 0 ? (uint32_t)a.mult[(int64_t)mt.cell * 16 + e] : 1u;
        return mt;
    };
    double2 pw2[8];
    double pxv = 0.0, px2 = 0.0, pxa = 0.0, psv = 0.0, pca = 0.0, pcb = 0.0;
    auto issue = [&](const Meta &mt) {
        const double2 *wr = reinterpret_cast<const double2 *>(lv.wcache + (size_t)mt.wsel * WAVE_WSTRIDE + cls * WAVE_ROW);
#pragma unroll
        for (int q = 0; q < 8; ++q) pw2[q] = wr[q];
        const int64_t o = (int64_t)mt.cell * ld + tc;
        pxv = xzero ? 0.0 : a.x[o];
        if (FUSED && a.x2) px2 = a.x2[o];
        if (FUSED && a.xacc)
            pxa = a.xacc[o];
        else if (FUSED && a.x3)
            pxa = a.x3[o];
        if (a.src) psv = a.src[o];
        if (FUSED && a.xcoarse) {
            pca = a.xcoarse[(int64_t)mt.cell * a.ldc + pa];
            pcb = a.xcoarse[(int64_t)mt.cell * a.ldc + pb];
        }
    };

    int64_t Q = blockIdx.x;
    if (Q >= nquad) return;
    Meta cur = meta_of(Q);
    issue(cur);
    Meta nxt = meta_of(Q + G < nquad ? Q + G : Q);

    for (;;) {
        const Meta me = cur;
        const int64_t o = (int64_t)me.cell * ld + t;
        const bool live = has && me.valid;
        double w[16];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            w[2 * q] = pw2[q].x;
            w[2 * q + 1] = pw2[q].y;
        }
        double val = pxv, rr = 0.0, pap = 0.0;
        if constexpr (FUSED) {
            if (a.xcoarse) {        // interpolate_and_sum_to!: identity rows += 1.0 c[a], midpoints += 0.5 c[a], += 0.5 c[b]
                const double ca = pca, cb = pcb;
                const double vi = val + ca;
                double vm = val;
                vm += 0.5 * ca;
                vm = vm + 0.5 * cb;
                val = pa == pb ? vi : vm;
            }
            if (a.xacc && live) a.xacc[o] = axpy1(ax, px2, pxa);
            if (a.x3) {
                const double t1 = axpy1(ax, px2, val);
                const double p2 = axpy1(beta, px2, pxa);
                val = axpy1(c2, p2, t1);
            } else if (a.x2)
                val = axpy1(beta, px2, val);
            if (a.xout && live) a.xout[o] = val;
            if (!has) val = 0.0;
            rr += val * val;
        }
        if (has) xs[lp] = val;
        const double sv = a.src ? psv : 0.0;
        // ---- the next quad's loads go out now; the words of the one after it are fetched here
        const bool more = Q + G < nquad;
        if (more) {
            issue(nxt);
            cur = nxt;
            nxt = meta_of(Q + 2 * G < nquad ? Q + 2 * G : Q + G);
        }
        // ---- evaluation: the lane's node with its class row in registers (stencil_eval_c's order)
        auto at = [&](int off) { return lds_ld(xs + max(L + off, 0)); };
        const double ctr = lds_ld(xs + L);
        double acc = w[0] * ctr;
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
        double out = sv + acc;
        if (cls > 0 && ((me.dm >> (cls - 1)) & 1u)) out = 0.0;
        if (live && (!FUSED || a.out)) a.out[o] = out;
        if constexpr (FUSED) {
            if (has) {
                const uint32_t mu = me.mw;
                pap += (double)mu * (ctr * out);
            }
            pap = dpp_add<0x111, 0xf>(pap);
            rr = dpp_add<0x111, 0xf>(rr);
            pap = dpp_add<0x112, 0xf>(pap);
            rr = dpp_add<0x112, 0xf>(rr);
            pap = dpp_add<0x114, 0xf>(pap);
            rr = dpp_add<0x114, 0xf>(rr);
            pap = dpp_add<0x118, 0xf>(pap);
            rr = dpp_add<0x118, 0xf>(rr);
            if (t == 15 && me.valid) {
                a.blockpart[2 * (int64_t)me.cell] = pap;
                a.blockpart[2 * (int64_t)me.cell + 1] = rr;
            }
        }
        if (!more) break;
        Q += G;
    }
}

bool apply_small_ok(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a, bool fused)
{
    if (!L.apply_small || L.apply_threads != 0) return false;
    if (!L.weight_cache || !lv.wcache || !mesh.cell_class) return false;
    if (lv.dim != 3 || lv.ncls != 15 || lv.nf > 192 || lv.nf < 4 || lv.m > 63 || !lv.pos32 || !lv.lpos) return false;
    if (a.flags & (2 | 8 | 16)) return false;                       // mass-only / driver-integral forms: weights not cached
    if (!(a.alpha == 1.0 || a.alpha == -1.0) || a.lambda != mesh.wc_lambda) return false;
    if (a.rcoarse) return false;
    if (!fused) return !a.xcoarse;
    if ((a.x3 || a.xacc) && !a.x2) return false;
    if (a.x3 && a.xacc) return false;
    if (a.xcoarse && (!lv.par32 || lv.nf_coarse < 1 || lv.nf_coarse > 64 || !a.xout || a.x2 || a.xacc || a.x3)) return false;
    return true;
}

void launch_apply_small(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a0, bool fused)
{
    ApplyArgs a = a0;
    const int64_t nblocks = a.cell_list ? a.ncell_list : a.ncells_prefix ? a.ncells_prefix : mesh.ncells;
    if (nblocks == 0) return;
    if (!a.x) throw std::runtime_error("operator apply: null input vector");
    if (!fused && !a.out) throw std::runtime_error("operator apply: a plain launch needs an output vector");
    if (fused && (!a.blockpart || !a.scal)) throw std::runtime_error("operator apply: fused launch without its reduction scratch");
    if ((a.flags & 1) && !mesh.dmask) throw std::runtime_error("operator apply: constraint requested without a Dirichlet mask");
    if ((a.flags & 128) && !(fused && a.x3 && a.x2 && a.xout && !a.xcoarse))
        throw std::runtime_error("operator apply: the zero-input form exists for the residual with two pending x-updates only");
    a.nwork = nblocks;
    if (lv.nf <= 16 && L.apply_pack && nblocks >= 4 && nblocks < (int64_t(1) << 31)) {
        const int64_t nquad = (nblocks + 3) / 4;
        const dim3 gq((unsigned)std::min<int64_t>(nquad, (int64_t)16 * L.num_cu)), bq(64);
        const size_t ldsq = sizeof(double) * (size_t)(4 * (lv.lds_g0 + lv.nf + lv.lds_g1) + 64);
        if (fused)
            hipLaunchKernelGGL((k_apply_pack<true>), gq, bq, ldsq, L.stream, lv, mesh.dmask, mesh.cell_class, a);
        else
            hipLaunchKernelGGL((k_apply_pack<false>), gq, bq, ldsq, L.stream, lv, mesh.dmask, mesh.cell_class, a);
        check_launch();
        if (L.n_small_launches) *L.n_small_launches += 1;
        return;
    }
    const int64_t grid = std::min<int64_t>(nblocks, (int64_t)(lv.nf > 64 ? 16 : 20) * L.num_cu);
    const size_t lds = sizeof(double) * (size_t)(SW + lv.lds_g0 + lv.nf + lv.lds_g1 + 64);
    const dim3 g((unsigned)grid), b(64);
    if (lv.nf > 64) {
        if (fused)
            hipLaunchKernelGGL((k_apply_small<3, true>), g, b, lds, L.stream, lv, mesh.dmask, mesh.cell_class, a);
        else
            hipLaunchKernelGGL((k_apply_small<3, false>), g, b, lds, L.stream, lv, mesh.dmask, mesh.cell_class, a);
    } else {
        if (fused)
            hipLaunchKernelGGL((k_apply_small<1, true>), g, b, lds, L.stream, lv, mesh.dmask, mesh.cell_class, a);
        else
            hipLaunchKernelGGL((k_apply_small<1, false>), g, b, lds, L.stream, lv, mesh.dmask, mesh.cell_class, a);
    }
    check_launch();
    if (L.n_small_launches) *L.n_small_launches += 1;
}

}  // namespace hmg
