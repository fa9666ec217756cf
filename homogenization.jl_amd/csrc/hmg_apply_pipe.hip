// k_apply_pipe: the operator apply of a 3D level whose cell fits the LDS twice (level 6: 6545 DOFs, 52 KB), as a
// software pipeline over cells.  ref: src/apply_local_operators.jl:85-133 (+ :7-27 residual, + constraint, + the CG
// pieces of src/multigrid.jl:54-68 when FUSED) -- the same arithmetic as k_apply (hmg_kernels.hip).
//
// One persistent 1024-thread workgroup per CU (~120 KB of LDS: two lattice images), 128 VGPRs per thread.  Per cell c:
//   A  issue the global loads of the NEXT cell into registers: its per-cell scalars (coefficients, Dirichlet mask,
//      multiplicities: one value per lane), the source values of the nodes this thread will evaluate (SRC), its
//      column(s) (x; FUSED: + x2, xacc / x3)
//   B  evaluate cell c from its LDS image: faces and edges one class per wave (SGPR weights, only the taps that
//      exist), corners by the last wave, cell interior register-blocked (6 nodes per thread, 56 LDS reads) -- results stored to HBM.  B reads
//      only registers and LDS: vmcnt counts in order, so a global load here would wait for the whole prefetch of A
//   C  the next cell's loads have landed meanwhile: FUSED element-wise work (p-update, x-update, r.r, folded
//      prolongation), scatter into the OTHER LDS image
//   D  one barrier
// so HBM loads are in flight while a cell is being evaluated, there is no workgroup launch / drain between cells,
// table words live in registers and the class table in LDS for the whole launch, and no weight table is shared
// between waves (each wave combines the two class rows it needs itself).  Cells are walked with a grid stride, so
// the cells in flight form one contiguous window of every vector.
#include "hmg_stencil.hpp"

#include <algorithm>

namespace hmg {

namespace {

constexpr int PIPE_NT = 1024, PIPE_SPT = 7, PIPE_NW = PIPE_NT / 64, PIPE_R = 6, PIPE_FI = 2, PIPE_RED = 4 * PIPE_NW,
              PIPE_CT = 15 * 15 * 7 + 1;   // LDS copy of the class table (padded to even)

#ifdef HMG_PHASE_TIMING   // dev build: thread 0 stamps the steps of every cell behind the 2 * ncells reduction partials
#define PIPE_STAMP(i)                                                                                   \
    do {                                                                                                \
        if (a.blockpart && !a.cell_list && tid == 0)                                                    \
            a.blockpart[2 * (size_t)a.nwork + 8 * cell + (i)] = (double)wall_clock64();                 \
    } while (0)
#else
#define PIPE_STAMP(i)
#endif

template <bool FUSED, bool SRC>
__global__ void __launch_bounds__(PIPE_NT, 4)   // 16 waves per CU: 4 per SIMD, 128 VGPRs
k_apply_pipe(LevelDev lv, const double *__restrict__ coef, const uint16_t *__restrict__ dmask, ApplyArgs a)
{
    constexpr int NT = PIPE_NT, SPT = PIPE_SPT, NW = PIPE_NW, R = PIPE_R, FI = PIPE_FI, NDIR = 15, NTERM = 7;
    constexpr int NSV = FI + 1 + R;          // source values a thread needs per cell: faces, edge, interior block (or corner)
    extern __shared__ double smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nf = lv.nf, m = lv.m;
    const int imgsz = (nf + lv.lds_g1 + 1) & ~1;
    double *W = smem;                        // class rows of the corners / edges: private to the last wave
    double *red = smem + WSZ;                // 2 (cell parity) x 2 x NW wave partials
    double *ctl = smem + WSZ + PIPE_RED;     // the level's class table (launch-invariant)
    double *img0 = ctl + PIPE_CT;            // two lattice images [nf | zero guard]
    double *cs = img0 + 2 * imgsz;           // coarse column of the cell being loaded (folded prolongation)

    // ---- launch-invariant per-thread tables ---------------------------------------------------------------------
    uint32_t lp2[(SPT + 1) / 2];             // LDS lattice position of the slots this thread loads, two per register
#pragma unroll
    for (int q = 0; q < SPT; q += 2) {
        const int t0 = tid + q * NT, t1 = tid + (q + 1) * NT;
        const uint32_t l0 = t0 < nf ? (uint32_t)lv.lpos[t0] : 0u;
        const uint32_t l1 = (q + 1 < SPT && t1 < nf) ? (uint32_t)lv.lpos[t1] : 0u;
        lp2[q / 2] = l0 | (l1 << 16);
    }
    const int face = wave >> 2, ft0 = (wave & 3) * (FI * 64);      // 4 waves per face, FI runs of 64 slots per wave
    const int fbase = lv.off_face + face * lv.nfi;
    // the 6 edges go to waves NW-6 .. NW-1 (one class per wave, like the faces), the 4 corners to the last wave
    const int edge = wave - (NW - 6), ebase = lv.off_edge + (edge < 0 ? 0 : edge) * lv.nei;
    uint32_t fw[FI], ew[1], cw = 0u;
#pragma unroll
    for (int q = 0; q < FI; ++q) {
        const int ti = ft0 + q * 64 + lane;
        fw[q] = ti < lv.nfi ? lv.pos32[fbase + ti] : 0u;
    }
    ew[0] = edge >= 0 && lane < lv.nei ? lv.pos32[ebase + lane] : 0u;
    if (wave == NW - 1 && lane < lv.ncorner) cw = lv.pos32[lane];
    uint32_t bw = tid < lv.nblk ? lv.blk_word[tid] : 0u;
    int bs = tid < lv.nblk ? (int)lv.blk_slot[tid] : 0;
    // per cell lanes 0..14 of every wave combine the interior row of the class table, lanes 16..30 the row of the
    // wave's face, lanes 32..46 the row of its edge
    for (int q = tid; q < lv.ncls * NDIR * NTERM; q += NT) ctl[q] = lv.ctab[q];
    const int cwrow = lane < 15 ? lane : (lane >= 16 && lane < 31) ? (1 + face) * NDIR + lane - 16
                      : (lane >= 32 && lane < 47 && edge >= 0) ? (1 + lv.nface + edge) * NDIR + lane - 32 : 0;
    for (int q = tid; q < imgsz - nf; q += NT) {
        img0[nf + q] = 0.0;
        img0[imgsz + nf + q] = 0.0;
    }

    // ---- launch-invariant scalars (moved to SGPRs explicitly: the VGPR budget belongs to the prefetched columns) ----
    const bool has_x2 = FUSED && a.x2, has_xout = FUSED && a.xout, has_xacc = FUSED && a.xacc, has_x3 = FUSED && a.x3;
    const bool has_cc = FUSED && a.xcoarse;
    const double beta = to_sgpr(has_x2 ? a.scal[a.s_num] / a.scal[a.s_den] : 0.0);
    const double ax = to_sgpr(has_xacc || has_x3 ? a.scal[a.a_num] / a.scal[a.a_den] : 0.0);
    const double c2 = to_sgpr(has_x3 ? a.scal[a.c_num] / a.scal[a.c_den] : 0.0);
    const int64_t ld = lv.ld;
    auto cell_of = [&](int64_t work) -> int64_t {
        if (!a.cell_list) return work;
        return (int64_t)__builtin_amdgcn_readfirstlane(a.cell_list[work]);
    };

    // ---- prefetch state of one cell ----------------------------------------------------------------------------------
    struct Pre {
        double cv;                 // lane t & 7: coef[t]
        uint32_t dmv, mv;          // Dirichlet mask; lane q & 3: multiplicity word q
        double sv[SRC ? NSV : 1];  // source values of the nodes this thread evaluates
    };
    double xv[SPT], x2v[FUSED ? SPT : 1], xav[FUSED ? SPT : 1], cval = 0.0;
    const int tlast = min(tid, nf - 1 - (SPT - 1) * NT);     // (host: (SPT - 1) * NT < nf <= SPT * NT)
    // (addresses are formed as uniform base + thread id, so that the loads take the scalar-base form)
    auto issue = [&](int64_t cell, Pre &p) {
        p.cv = (coef + cell * 8)[lane & 7];
        p.dmv = (a.flags & 1) ? (uint32_t)dmask[cell] : 0u;
        p.mv = FUSED ? (reinterpret_cast<const uint32_t *>(a.mult) + cell * 4)[lane & 3] : 0u;
        if (SRC) {
            const double *sc = a.src + cell * ld;
#pragma unroll
            for (int q = 0; q < FI; ++q) p.sv[q] = sc[fbase + min(ft0 + q * 64 + lane, lv.nfi - 1)];
            p.sv[FI] = sc[ebase + min(lane, lv.nei - 1)];
            // (one straight-line sequence for all waves: surplus loads re-read a slot the thread reads anyway)
            int slot[R];
            block_slots<R>(m, bw, bs, slot);
            const int nv = (int)(bw >> 28);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int t = wave == NW - 1 ? min(lane, lv.ncorner - 1) : (r < nv ? slot[r] : bs);
                p.sv[FI + 1 + r] = sc[t];
            }
        }
        // (straight-line, unpredicated: the last chunk's surplus lanes re-read its last slot.  Loads under per-lane
        //  branches end up one basic block each and the backend then serialises them with vmcnt waits)
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int64_t off = cell * ld + q * NT;
            const int tq = q < SPT - 1 ? tid : tlast;
            xv[q] = (a.x + off)[tq];
            if (FUSED) {
                x2v[q] = has_x2 ? (a.x2 + off)[tq] : 0.0;
                xav[q] = has_xacc ? (a.xacc + off)[tq] : has_x3 ? (a.x3 + off)[tq] : 0.0;
            }
        }
        if (FUSED && has_cc && tid < lv.nf_coarse) cval = (a.xcoarse + cell * a.ldc)[tid];
    };
    auto finish = [&](int64_t cell, double *img, double &rr) {
        uint32_t pw[FUSED ? SPT : 1];
        if (FUSED && has_cc) {               // coarse column complete before the first use
#pragma unroll
            for (int q = 0; q < SPT; ++q) pw[q] = tid + q * NT < nf ? (lv.par32 + q * NT)[tid] : 0u;
            for (int q = tid + NT; q < lv.nf_coarse; q += NT) cs[q] = a.xcoarse[cell * a.ldc + q];
            if (tid < lv.nf_coarse) cs[tid] = cval;
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int64_t off = cell * ld + q * NT;
            if (tid + q * NT < nf) {
                double v = xv[q];
                if (FUSED) {
                    if (has_cc) {            // interpolate_and_sum_to! (src/interpolation.jl:64-74), CSC column order
                        const uint32_t pa = pw[q] & 0xffffu, pb = pw[q] >> 16;
                        if (pa == pb)
                            v = v + cs[pa];
                        else {
                            v += 0.5 * cs[pa];
                            v = v + 0.5 * cs[pb];
                        }
                    }
                    if (has_xacc) (a.xacc + off)[tid] = xav[q] + ax * x2v[q];
                    if (has_x3) {            // two pending CG x-updates, same three roundings as done one by one
                        const double t1 = v + ax * x2v[q];
                        const double p2 = xav[q] + beta * x2v[q];
                        v = t1 + c2 * p2;
                    } else if (has_x2)
                        v = v + beta * x2v[q];
                    if (has_xout) (a.xout + off)[tid] = v;
                    rr += v * v;
                }
                img[(lp2[q / 2] >> (16 * (q & 1))) & 0xffffu] = v;
            }
        }
    };

    const int64_t stride = gridDim.x;
    int64_t work = blockIdx.x;
    if (work >= a.nwork) return;
    Pre cur, nxt;
    double rr_cur = 0.0;
    {
        const int64_t cell = cell_of(work);
        issue(cell, cur);
        finish(cell, img0, rr_cur);
        __syncthreads();                     // (also: class table and guard zones written)
    }
    nxt = cur;
    int par = 0;
    for (; work < a.nwork; work += stride, par ^= 1) {
        const int64_t cell = cell_of(work);
        const bool has_next = work + stride < a.nwork;
        const int64_t cell_n = has_next ? cell_of(work + stride) : cell;
        // (the table words pass through an opaque move once per cell: everything decoded from them -- dozens of LDS
        //  addresses and slots -- would otherwise be hoisted out of the cell loop as loop-invariant and spilled)
        asm volatile("" : "+v"(bw), "+v"(bs), "+v"(fw[0]), "+v"(fw[1]), "+v"(ew[0]), "+v"(cw));
        PIPE_STAMP(0);
        // ---- A ------------------------------------------------------------------------------------------------
        if (has_next) issue(cell_n, nxt);
        PIPE_STAMP(1);
        // ---- B ------------------------------------------------------------------------------------------------
        const double *xs = img0 + par * imgsz;
        const double *sc = SRC ? a.src + cell * ld : nullptr;
        double *oc = a.out ? a.out + cell * ld : nullptr;
        double pap = 0.0;
        double s[NTERM];
#pragma unroll
        for (int t = 0; t < NTERM - 1; ++t) s[t] = (a.flags & 2) ? 0.0 : a.alpha * readlane_f64(cur.cv, t);
        s[NTERM - 1] = a.alpha * a.lambda * readlane_f64(cur.cv, NTERM - 1);
        const uint32_t dmu = __builtin_amdgcn_readfirstlane(cur.dmv);
        uint32_t mqu[4] = {0u, 0u, 0u, 0u};
        if (FUSED) {
#pragma unroll
            for (int q = 0; q < 4; ++q) mqu[q] = __builtin_amdgcn_readlane(cur.mv, q);
        }
        double wv = 0.0;                      // lane d: interior weight d; lane 16 + d: weight d of this wave's face
#pragma unroll
        for (int t = 0; t < NTERM; ++t) wv += ctl[cwrow * NTERM + t] * s[t];
        {
            const bool fdir = (dmu >> face) & 1u;
            const double fmult = (double)((mqu[0] >> (8 * face)) & 0xffu);
            double pre[FI];
#pragma unroll
            for (int q = 0; q < FI; ++q) pre[q] = SRC ? cur.sv[q] : 0.0;
            if (face == 0)
                face_items<0, FI, FUSED>(wv, 16, xs, m, lv.nfi, fbase, ft0, fw, fdir, fmult, sc, oc, pap, lane, pre, true);
            else if (face == 1)
                face_items<1, FI, FUSED>(wv, 16, xs, m, lv.nfi, fbase, ft0, fw, fdir, fmult, sc, oc, pap, lane, pre, true);
            else if (face == 2)
                face_items<2, FI, FUSED>(wv, 16, xs, m, lv.nfi, fbase, ft0, fw, fdir, fmult, sc, oc, pap, lane, pre, true);
            else
                face_items<3, FI, FUSED>(wv, 16, xs, m, lv.nfi, fbase, ft0, fw, fdir, fmult, sc, oc, pap, lane, pre, true);
        }
        if (edge >= 0) {                      // this wave's edge: one class, SGPR weights, only the taps that exist
            const bool edir = (dmu >> (lv.nface + edge)) & 1u;
            const int eb = lv.nface + edge;
            const double emult = (double)(((eb < 4 ? mqu[0] : eb < 8 ? mqu[1] : mqu[2]) >> (8 * (eb & 3))) & 0xffu);
            double pre[1] = {SRC ? cur.sv[FI] : 0.0};
            switch (edge) {
            case 0: class_items<edge_tap_mask(0), 1, FUSED>(wv, 32, xs, m, lv.nei, ebase, 0, ew, edir, emult, sc, oc, pap, lane, pre, true); break;
            case 1: class_items<edge_tap_mask(1), 1, FUSED>(wv, 32, xs, m, lv.nei, ebase, 0, ew, edir, emult, sc, oc, pap, lane, pre, true); break;
            case 2: class_items<edge_tap_mask(2), 1, FUSED>(wv, 32, xs, m, lv.nei, ebase, 0, ew, edir, emult, sc, oc, pap, lane, pre, true); break;
            case 3: class_items<edge_tap_mask(3), 1, FUSED>(wv, 32, xs, m, lv.nei, ebase, 0, ew, edir, emult, sc, oc, pap, lane, pre, true); break;
            case 4: class_items<edge_tap_mask(4), 1, FUSED>(wv, 32, xs, m, lv.nei, ebase, 0, ew, edir, emult, sc, oc, pap, lane, pre, true); break;
            default: class_items<edge_tap_mask(5), 1, FUSED>(wv, 32, xs, m, lv.nei, ebase, 0, ew, edir, emult, sc, oc, pap, lane, pre, true); break;
            }
        }
        if (wave == NW - 1) {
            // the 4 corners (4 classes): this wave builds their class rows for itself (LDS operations of one wave
            // execute in order: no barrier between the writes and the reads below) and evaluates them with per-lane
            // weights
            const int first = (1 + lv.nface + lv.nedge) * NDIR, last = lv.ncls * NDIR;
            for (int idx = first + lane; idx < last; idx += 64) {
                double w = 0.0;
#pragma unroll
                for (int t = 0; t < NTERM; ++t) w += ctl[idx * NTERM + t] * s[t];
                W[idx] = w;
            }
            if (lane < lv.ncorner) {
                const int t = lane;
                const double sv = SRC ? cur.sv[FI + 1] : 0.0;
                int L, len, A, B, cls;
                decode32<3>(cw, m, L, len, A, B, cls);
                double ctr;
                double o = sv + stencil_eval_c<3>(W + cls * NDIR, xs, L, len, A, B, ctr);
                if ((dmu >> (cls - 1)) & 1u) o = 0.0;
                if (!FUSED || oc) oc[t] = o;
                if (FUSED) {
                    const int e = cls - 1;
                    const uint32_t word = e < 4 ? mqu[0] : e < 8 ? mqu[1] : e < 12 ? mqu[2] : mqu[3];
                    pap += (double)((word >> (8 * (e & 3))) & 0xffu) * (ctr * o);
                }
            }
        } else if (tid < lv.nblk) {
            double w0[15];
#pragma unroll
            for (int d = 0; d < NDIR; ++d) w0[d] = readlane_f64(wv, d);
            double acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = SRC ? cur.sv[FI + 1 + r] : 0.0;
            interior_block_core<R, FUSED>(w0, xs, m, nf >> 1, bw, bs, oc, pap, acc);
        }
        PIPE_STAMP(2);
        // ---- C ------------------------------------------------------------------------------------------------
        double rr_next = 0.0;
        if (has_next) finish(cell_n, img0 + (par ^ 1) * imgsz, rr_next);
        PIPE_STAMP(3);
        // ---- D ------------------------------------------------------------------------------------------------
        if (FUSED) {
            const double wp = wave_sum(pap), wr = wave_sum(rr_cur);
            if (lane == 0) {
                red[par * 2 * NW + wave] = wp;
                red[par * 2 * NW + NW + wave] = wr;
            }
        }
        __syncthreads();
        PIPE_STAMP(4);
        if (FUSED && tid == 0) {              // (the other parity's partials are written before the NEXT barrier)
            double sp = 0.0, sr = 0.0;
            for (int q = 0; q < NW; ++q) {
                sp += red[par * 2 * NW + q];
                sr += red[par * 2 * NW + NW + q];
            }
            a.blockpart[2 * cell] = sp;
            a.blockpart[2 * cell + 1] = sr;
        }
        rr_cur = rr_next;
        cur = nxt;
    }
}

}  // namespace

size_t apply_pipe_lds_bytes(const LevelDev &lv, bool with_coarse)
{
    const int imgsz = (lv.nf + lv.lds_g1 + 1) & ~1;
    return sizeof(double) * (size_t)(WSZ + PIPE_RED + PIPE_CT + 2 * imgsz + (with_coarse ? lv.nf_coarse : 0));
}

// Can the pipelined kernel run this level?  (3D, register-block tables for R = 6 that fit waves 0..14, face / edge
// run counts the kernel is compiled for, the class table, both images + the coarse column within the LDS of one CU,
// and more than half of it: the grid is one workgroup per CU)
bool apply_pipe_supported(const LevelDev &lv)
{
    return lv.dim == 3 && lv.blk_R == PIPE_R && lv.nblk <= (PIPE_NW - 1) * 64 && lv.nf <= PIPE_SPT * PIPE_NT &&
           lv.nf > (PIPE_SPT - 1) * PIPE_NT &&
           lv.nfi <= 4 * PIPE_FI * 64 && lv.nei <= 64 && lv.nedge == 6 && lv.ncorner == 4 && lv.nface == 4 &&
           lv.nf_coarse <= 2 * PIPE_NT && lv.ncls == 15 &&
           lv.nterm == 7 && apply_pipe_lds_bytes(lv, true) <= 160 * 1024 && apply_pipe_lds_bytes(lv, false) > 80 * 1024;
}

template <bool FUSED, bool SRC>
static void launch_pipe(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a, size_t lds, unsigned grid)
{
    auto kern = k_apply_pipe<FUSED, SRC>;
    HMG_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PIPE_NT), lds, L.stream, lv, mesh.coef, mesh.dmask, a);
    check_launch();
}

// fused launches with a source vector (the two folded residuals of hmg_vcycle) would hold four prefetched streams:
// they stay with k_apply
bool apply_pipe_takes(const LevelDev &lv, const ApplyArgs &a, bool fused)
{
    return apply_pipe_supported(lv) && !(fused && a.src);
}

void launch_apply_pipe(const Launch &L, const LevelDev &lv, const MeshDev &mesh, ApplyArgs a, bool fused)
{
    const int64_t nwork = a.cell_list ? a.ncell_list : mesh.ncells;
    if (nwork == 0) return;
    a.nwork = nwork;
    const size_t lds = apply_pipe_lds_bytes(lv, fused && a.xcoarse);
    const unsigned grid = (unsigned)std::min<int64_t>(nwork, L.num_cu);
    if (fused && a.src) throw std::runtime_error("pipelined apply: fused launch with a source vector");
    if (fused)
        launch_pipe<true, false>(L, lv, mesh, a, lds, grid);
    else if (a.src)
        launch_pipe<false, true>(L, lv, mesh, a, lds, grid);
    else
        launch_pipe<false, false>(L, lv, mesh, a, lds, grid);
}

}  // namespace hmg
