// Device-side building blocks of the kernels of hmg_kernels.hip: error checking,
// reductions, LDS reads, addressing-word decoders, stencil evaluators (per node, per face run, register-blocked).
#pragma once
#include "hmg_device.hpp"

#include <cstdio>
#include <stdexcept>
#include <string>

namespace hmg {

#define HMG_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + \
                                     __FILE__ + ":" + std::to_string(__LINE__));                \
    } while (0)

static inline void check_launch() { HMG_HIP_CHECK(hipGetLastError()); }

constexpr int WSZ = 232;   // LDS doubles reserved for the class weight table (>= 15*15, 16-B multiple)
constexpr int WSZ_RB = 64; // ... of the register-blocked instantiations: only the corners' rows (4 x 15) live in LDS.  With it a
                           // level-6 cell takes 53 192 B: three workgroups per CU (3 x 53 248 <= 160 KB)

// uniform (wave-invariant) loads through the constant address space become scalar loads (s_load_*): results land in SGPRs
#define HMG_KP(T, p) ((const T __attribute__((address_space(4))) *)(uintptr_t)(p))

// The vector updates of the CG smoother (x += alpha p, p = r + beta p, r -= alpha q) are ONE rounding each, everywhere: the same
// update is done by different kernels depending on which folds are on (the load phase of an apply, a streaming kernel, both
// pending updates at once), and the library promises the same bits from all of them.  Left to the backend's contraction, a
// product hoisted in front of a test of an optional pointer became a separate multiply in some instantiations (round 4: the
// one-wave level-4 kernel rounded x += alpha p twice, k_cg_xp_update once).
__device__ __forceinline__ double axpy1(double a, double x, double y) { return __builtin_fma(a, x, y); }   // a x + y

// A global store the compiler does not COUNT.  In a loop that requests the next item's loads before it stores the current item's
// results, a compiler-visible store between a load and its first use on SOME control-flow path turns the wait for that load into a
// vmcnt(0) -- the store's round trip to L2 (~2000 cycles) per item (measured on the level-7 kernel, profiles/r05_level7_phase_timing.txt).
// As an asm statement the store is invisible to that bookkeeping: the waits see loads only and stay exact, or err towards one more
// LOAD.  The data registers must outlive the issue: s_nop 1 (guide section 5.7).
__device__ __forceinline__ void st_global(double *p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v));
}

// two consecutive doubles in one store instruction (8-byte aligned address).  A store-heavy tail is bound by the number of store
// INSTRUCTIONS a CU can issue, not by their bytes (MI355X_MICROARCH.md, "store-ISSUE-bound"): 16 bytes per lane halve it.
typedef double hmg_f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_global2(double *p, double lo, double hi)
{
    hmg_f64x2 v;
    v.x = lo;
    v.y = hi;
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v));
}

// the value of the other lane of this lane's pair (lanes 2 i and 2 i + 1)
__device__ __forceinline__ double pair_swap_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, false);   // quad_perm [1, 0, 3, 2]
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// The same sum by data-parallel-primitive moves (row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then row_bcast 15 / 31 across the
// rows): no LDS round trip per level -- wave_sum's __shfl_down is a ds_bpermute, six dependent ones per sum, which for the one-wave
// kernels of the small levels was a quarter of a cell's life.  The total arrives in LANE 63.  (Another summation order than
// wave_sum: used where only one kernel ever forms the sum.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    // lanes the move does not reach keep `old` = 0: +0.0 is added there
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum63(double v)
{
    v = dpp_add<0x111, 0xf>(v);     // row_shr:1
    v = dpp_add<0x112, 0xf>(v);     // row_shr:2
    v = dpp_add<0x114, 0xf>(v);     // row_shr:4
    v = dpp_add<0x118, 0xf>(v);     // row_shr:8   -> lane 15 of every row holds the row's sum
    v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3  -> lane 63 holds the wave's sum
    return v;
}

// all threads of the block call; result valid in thread 0. red: >= blockDim/64 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *red)
{
    v = wave_sum63(v);                  // (round 4: cross-lane moves instead of six LDS round trips; the wave's sum is in lane 63)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) red[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) s += red[i];
    }
    __syncthreads();
    return s;
}

// Sum of nb block partials, computed redundantly (same order, same value) by every block that needs the scalar:
// saves the separate finalize launch in launch-bound loops.  All threads call; result valid in all threads.
__device__ __forceinline__ double sum_partials_all(const double *__restrict__ part, int nb, double *red, double *bc)
{
    double a = 0.0;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) a += part[i];
    const double s = block_sum(a, red);
    if (threadIdx.x == 0) *bc = s;
    __syncthreads();
    return *bc;
}


// Per-cell scale of every operator term: alpha*|J|*P_kl for the diffusion terms, alpha*lambda*|J| for the
// mass term; mass_only (next_rhs!) zeroes the diffusion part.
// flags bit 1: mass term only; bit 4: ... and not scaled by |J| (the driver integrals scale the per-cell sums instead)
template <int DIM>
__device__ __forceinline__ void cell_scales(const double *__restrict__ cc, double alpha, double lambda, double *s,
                                            int flags = 0)
{
    constexpr int NTERM = DIM == 3 ? 7 : 4;
#pragma unroll
    for (int t = 0; t < NTERM - 1; ++t) s[t] = (flags & 2) ? 0.0 : alpha * cc[t];
    s[NTERM - 1] = alpha * lambda * ((flags & 16) ? 1.0 : cc[NTERM - 1]);
}

// LDS read that the backend must not fuse into ds_read2_b64 (8 LDS cycles for 2x8 B per lane, half
// the rate of two ds_read_b64 -- MI355X_MICROARCH.md, LDS table).
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ double lds_ld(const double *p)
{
    return *(const volatile lds_f64 *)(const lds_f64 *)p;   // stays a ds_read_b64 (LDS address space)
}

template <int DIM>
__device__ __forceinline__ double stencil_eval_v(const double *w, const double *p, int len, int A, int B, double &ctr)
{
    ctr = lds_ld(p);
    double acc = w[0] * ctr;
    acc += w[1] * lds_ld(p + 1);
    acc += w[2] * lds_ld(p - 1);
    acc += w[3] * lds_ld(p + len - 1);
    acc += w[4] * lds_ld(p - len);
    acc += w[5] * lds_ld(p + len);
    acc += w[6] * lds_ld(p - len - 1);
    if (DIM == 3) {
        const double *pu = p + A;
        const double *pd = p - B;
        acc += w[7] * lds_ld(pu - len);
        acc += w[8] * lds_ld(pd + len + 1);
        acc += w[9] * lds_ld(pu - 1);
        acc += w[10] * lds_ld(pd + 1);
        acc += w[11] * lds_ld(pu);
        acc += w[12] * lds_ld(pd);
        acc += w[13] * lds_ld(pu + 1 - len);
        acc += w[14] * lds_ld(pd + len);
    }
    return acc;
}

// Surface nodes: the weight row comes from the LDS class table, read tap by tap (volatile LDS reads keep
// this order, so only a few registers are live); zero-weight taps may address below the lattice image
// (plane -1 / row -1) and are clamped to 0.
template <int DIM>
__device__ __forceinline__ double stencil_eval_c(const double *wr, const double *base, int L, int len, int A, int B,
                                                 double &ctr)
{
    auto at = [&](int off) { return lds_ld(base + max(L + off, 0)); };
    ctr = lds_ld(base + L);
    double acc = lds_ld(wr + 0) * ctr;
    acc += lds_ld(wr + 1) * lds_ld(base + L + 1);
    acc += lds_ld(wr + 2) * at(-1);
    acc += lds_ld(wr + 3) * lds_ld(base + L + len - 1);
    acc += lds_ld(wr + 4) * at(-len);
    acc += lds_ld(wr + 5) * lds_ld(base + L + len);
    acc += lds_ld(wr + 6) * at(-len - 1);
    if (DIM == 3) {
        acc += lds_ld(wr + 7) * lds_ld(base + L + A - len);
        acc += lds_ld(wr + 8) * at(len + 1 - B);
        acc += lds_ld(wr + 9) * lds_ld(base + L + A - 1);
        acc += lds_ld(wr + 10) * at(1 - B);
        acc += lds_ld(wr + 11) * lds_ld(base + L + A);
        acc += lds_ld(wr + 12) * at(-B);
        acc += lds_ld(wr + 13) * lds_ld(base + L + A + 1 - len);
        acc += lds_ld(wr + 14) * at(len - B);
    }
    return acc;
}

__device__ __forceinline__ double to_sgpr(double v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const unsigned hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}

// Compact addressing word of the LDS-resident kernels: L | j<<16 | k<<22 | cls<<28 in 3D (j, k <= 63, i.e.
// levels whose cell fits the LDS), L | j<<16 | cls<<28 in 2D (j <= 255).  Rows have len = m+1-j-k nodes;
// A / B are the address offsets to the same (i,j) in the planes above / below.
template <int DIM>
__device__ __forceinline__ void decode32(uint32_t w, int m, int &L, int &len, int &A, int &B, int &cls)
{
    L = (int)(w & 0xffffu);
    cls = (int)(w >> 28);
    if (DIM == 3) {
        const int j = (int)((w >> 16) & 63u), k = (int)((w >> 22) & 63u);
        len = m + 1 - j - k;
        const int n = m - k;
        const int Tk = (int)(__umul24((uint32_t)n + 1u, (uint32_t)n + 2u) >> 1);   // n <= 63: full-rate 24-bit multiply
        A = Tk - j;
        B = Tk + n + 2 - j;
    } else {
        const int j = (int)((w >> 16) & 255u);
        len = m + 1 - j;
        A = 0;
        B = 0;
    }
}

// Wide word of the slab kernel (cells larger than the LDS): i | j<<7 | k<<14 | cls<<21, L recomputed.
__device__ __forceinline__ void decode32w(uint32_t w, int m, int &L, int &len, int &A, int &B, int &cls, int &k)
{
    const int i = (int)(w & 127u), j = (int)((w >> 7) & 127u);
    k = (int)((w >> 14) & 127u);
    cls = (int)(w >> 21);
    len = m + 1 - j - k;
    const int n = m - k;
    const int Tk = ((n + 1) * (n + 2)) >> 1;
    A = Tk - j;
    B = Tk + n + 2 - j;
    const int full = (m + 1) * (m + 2) * (m + 3), rest = (n + 1) * (n + 2) * (n + 3);
    L = (full - rest) / 6 + j * (n + 1) - ((j * (j - 1)) >> 1) + i;
}

// Interior entries of the slab kernel's evaluation lists (round 4): the lattice position comes with the word, L | j << 16 | k << 23
// -- decode32w spends a dozen integer instructions per node on it (two tetrahedral numbers, a division by 6).
__device__ __forceinline__ void decode_lattice(uint32_t w, int m, int &L, int &len, int &A, int &B)
{
    L = (int)(w & 0xffffu);
    const int j = (int)((w >> 16) & 127u), k = (int)((w >> 23) & 127u);
    len = m + 1 - j - k;
    const int n = m - k;
    const int Tk = ((n + 1) * (n + 2)) >> 1;
    A = Tk - j;
    B = Tk + n + 2 - j;
}

// Stencil taps that leave the cell at a node in the interior of face F (F = 0: k = 0, 1: j = 0, 2: i = 0, 3: i+j+k = m;
// tap numbering of stencil_eval_v): their class weights are zero, the host checks that against the class table.
__host__ __device__ constexpr uint32_t face_tap_mask(int f)
{
    return 0x7fffu & ~(f == 0   ? (1u << 8 | 1u << 10 | 1u << 12 | 1u << 14)
                       : f == 1 ? (1u << 4 | 1u << 6 | 1u << 7 | 1u << 13)
                       : f == 2 ? (1u << 2 | 1u << 3 | 1u << 9 | 1u << 14)
                                : (1u << 1 | 1u << 5 | 1u << 11 | 1u << 13));
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const unsigned lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const unsigned hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double((int)hi, (int)lo);
}

// Face-interior nodes, one entity class per wave: the class weight row sits in SGPRs (one LDS read + readlanes per
// wave instead of 15 LDS reads per node) and only the 11 taps that exist are read -- 11 LDS reads per node instead
// of 30.  A wave evaluates up to NITEM runs of 64 consecutive slots of face F; fw[] holds their addressing words.
template <uint32_t M, int NITEM, bool FUSED, bool PAP = FUSED>
__device__ __forceinline__ void class_items(double wv, int wlane0, const double *xs, int m, int nfi, int slot_base, int t0,
                                            const uint32_t (&fw)[NITEM], bool dirichlet, double mult, const double *sc,
                                            double *oc, double &pap, int lane, const double (&pre)[NITEM], bool use_pre,
                                            bool wdot = false, double *keep = nullptr)
{
    // keep (optional, NITEM entries): the results are handed back as well (restriction in the epilogue)
    // wdot: the source value multiplies instead of being added: out = alpha A x, pap += mult (x + src) out
    // wv: the class weight row, spread over the lanes wlane0 .. wlane0 + 14 of this wave; M: the taps that exist
    double w[15];
#pragma unroll
    for (int d = 0; d < 15; ++d) w[d] = (M >> d) & 1u ? readlane_f64(wv, wlane0 + d) : 0.0;
#pragma unroll
    for (int q = 0; q < NITEM; ++q) {
        const int ti = t0 + q * 64 + lane;
        if (ti < nfi) {
            const int t = slot_base + ti;
            double o = 0.0;
            if (!dirichlet) {
                const double sv = use_pre ? pre[q] : sc ? sc[t] : 0.0;
                int L, len, A, B, cls;
                decode32<3>(fw[q], m, L, len, A, B, cls);
                const double *p = xs + L;
                const double ctr = lds_ld(p);
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
                o = wdot ? acc : sv + acc;
                if (PAP) pap += mult * ((wdot ? ctr + sv : ctr) * o);
            }
            if (!FUSED || oc) oc[t] = o;
            if (keep) keep[q] = o;
        }
    }
}

template <int F, int NITEM, bool FUSED, bool PAP = FUSED>
__device__ __forceinline__ void face_items(double wv, int wlane0, const double *xs, int m, int nfi, int slot_base, int t0,
                                           const uint32_t (&fw)[NITEM], bool dirichlet, double mult, const double *sc,
                                           double *oc, double &pap, int lane, const double (&pre)[NITEM], bool use_pre,
                                           bool wdot = false, double *keep = nullptr)
{
    class_items<face_tap_mask(F), NITEM, FUSED, PAP>(wv, wlane0, xs, m, nfi, slot_base, t0, fw, dirichlet, mult, sc, oc, pap, lane,
                                                     pre, use_pre, wdot, keep);
}

// Edge e of the reference simplex lies on two faces (edge order of the reference: (1,2) (1,3) (1,4) (2,3) (2,4) (3,4) =
// f0&f1 f0&f2 f1&f2 f0&f3 f1&f3 f2&f3): a node in its interior keeps the taps both faces keep.
__host__ __device__ constexpr uint32_t edge_tap_mask(int e)
{
    return face_tap_mask(e == 0 || e == 1 || e == 3 ? 0 : e == 2 || e == 4 ? 1 : 2) &
           face_tap_mask(e == 0 ? 1 : e == 1 || e == 2 ? 2 : 3);
}

// Register-blocked evaluation of the cell interior (3D).  A thread owns R nodes (i, j, k0 .. k0+R-1) -- the same (i,j)
// in R consecutive k-planes -- and walks the planes k0-1 .. k0+R once: in every plane it reads the 7 lattice points
// (i,j-1) (i+1,j-1) | (i-1,j) (i,j) (i+1,j) | (i-1,j+1) (i,j+1) and feeds each value to every node of the thread
// that taps it (the node in that plane takes all 7, the node below the first 4, the node above the last 4):
// 7R + 14 LDS reads for R nodes instead of 15R, one addressing word per R nodes, R independent FMA chains.  Lanes
// enumerate the interior (i,j) of plane k0 in lattice order, so a wave reads (nearly) consecutive LDS words.
// Block word: L(i,j,k0) | j << 16 | k0 << 22 | nv << 28, nv = number of the R nodes that exist (i+j+k <= m-1);
// blk_slot: storage slot of node (i,j,k0).  Planes that only non-existent nodes would tap are read at a safe
// address in the middle of the image (values unused).
// Storage slots of the R nodes of a block: the cell interior is itself a lattice (m' = m - 4) stored in lattice order,
// slot(r+1) - slot(r) = T(n0 - 3 - r) - (j - 1), n0 = m - k0.
template <int R>
__device__ __forceinline__ void block_slots(int m, uint32_t word, int slot0, int (&slot)[R])
{
    const int j = (int)((word >> 16) & 63u), n0 = m - (int)((word >> 22) & 63u);
    int sl = slot0, ds = (int)(__umul24((uint32_t)(n0 - 2), (uint32_t)(n0 - 1)) >> 1) - j + 1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        slot[r] = sl;
        sl += ds;
        ds -= n0 - 2 - r;
    }
}

// acc[r] holds the value the sum of node r starts from (0, or the source value for out = src + alpha A x).
// WDOT: wsv[r] is added to the node's own value in the p.Ap-type sum (driver integrals: (v + w) . M v).
template <int R, bool FUSED, bool WDOT = false, bool KEEP = false>
__device__ __forceinline__ void interior_block_core(const double (&w)[15], const double *xs, int m, int safe, uint32_t word,
                                                    int slot0, double *oc, double &pap, double (&acc)[R],
                                                    const double *wsv = nullptr)
{
    // KEEP: the finished sums stay in acc[] for the caller (restriction in the epilogue) and p.Ap is not formed
    const int L = (int)(word & 0xffffu), j = (int)((word >> 16) & 63u), k0 = (int)((word >> 22) & 63u);
    const int nv = (int)(word >> 28);
    const int n0 = m - k0;
    const int ds0 = (int)(__umul24((uint32_t)(n0 - 2), (uint32_t)(n0 - 1)) >> 1) - j + 1;
    double ctr[R];
    int len = n0 + 2 - j;                                                                  // row j of plane k0-1
    int delta = (int)(__umul24((uint32_t)(n0 + 2), (uint32_t)(n0 + 3)) >> 1) - j;         // q(k0) - q(k0-1) = T(n0+1) - j
    int q = L - delta;
    int slot = slot0, ds = ds0;
#pragma unroll
    for (int s = 0; s < R + 2; ++s) {
        const int qs = s <= nv + 1 ? q : safe;
        const double *a1 = xs + (qs - len - 1), *a2 = xs + (qs - 1), *a3 = xs + (qs + len - 1);
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3, v4 = 0.0, v5 = 0.0, v6 = 0.0;
        if (s >= 1) {
            v0 = lds_ld(a1);
            v1 = lds_ld(a1 + 1);
            v2 = lds_ld(a2);
        }
        v3 = lds_ld(a2 + 1);
        if (s <= R) {
            v4 = lds_ld(a2 + 2);
            v5 = lds_ld(a3);
            v6 = lds_ld(a3 + 1);
        }
        if (s >= 2) {                       // upper plane of node s-2: its last taps, then it is done
            double t = acc[s - 2];
            t += w[7] * v0;
            t += w[13] * v1;
            t += w[9] * v2;
            t += w[11] * v3;
            if (KEEP) acc[s - 2] = t;
            if (s - 2 < nv) {
                if (!FUSED || oc) oc[slot] = t;
                if (FUSED && !KEEP) pap += (WDOT ? ctr[s - 2] + wsv[s - 2] : ctr[s - 2]) * t;   // (the CG passes have no src)
            }
            slot += ds;
            ds -= n0 - s;                   // n0 - 2 - r, r = s - 2
        }
        if (s >= 1 && s <= R) {             // the plane of node s-1
            double t = acc[s - 1];
            if (!KEEP) ctr[s - 1] = v3;
            t += w[0] * v3;
            t += w[1] * v4;
            t += w[2] * v2;
            t += w[3] * v5;
            t += w[4] * v1;
            t += w[5] * v6;
            t += w[6] * v0;
            acc[s - 1] = t;
        }
        if (s < R) {                        // lower plane of node s
            double t = acc[s];
            t += w[12] * v3;
            t += w[10] * v4;
            t += w[14] * v5;
            t += w[8] * v6;
            acc[s] = t;
        }
        q += delta;
        delta -= n0 + 2 - s;               // T(n) - T(n-1) = n + 1, n = n0 + 1 - s
        len -= 1;
        // (scheduling fence: without it the backend hoists the LDS reads of several planes above the FMAs of the
        //  first and spills; one plane's 7 values + 3 running sums is what has to be live -- 64-VGPR budget)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// SRC: out = src + alpha A x -- the R source values are loaded first (their global latency hides behind the planes).
// The LDS lattice positions of the R nodes of a block (the recurrence of interior_block_core).
template <int R>
__device__ __forceinline__ void block_positions(int m, uint32_t word, int (&pos)[R])
{
    const int L = (int)(word & 0xffffu), j = (int)((word >> 16) & 63u), n0 = m - (int)((word >> 22) & 63u);
    int delta = (int)(__umul24((uint32_t)(n0 + 2), (uint32_t)(n0 + 3)) >> 1) - j;
    int q = L - delta;
#pragma unroll
    for (int s = 0; s < R + 1; ++s) {
        if (s >= 1) pos[s - 1] = q;
        q += delta;
        delta -= n0 + 2 - s;
    }
}

// ... with the results handed back in res[] (KEEP form: out = src + alpha A x, no p.Ap)
template <int R, bool FUSED>
__device__ __forceinline__ void interior_block_keep(const double (&w)[15], const double *xs, int m, int safe, uint32_t word,
                                                    int slot0, const double *sc, double *oc, double (&res)[R])
{
    if (sc) {
        int slot[R];
        block_slots<R>(m, word, slot0, slot);
        const int nv = (int)(word >> 28);
#pragma unroll
        for (int r = 0; r < R; ++r) res[r] = r < nv ? sc[slot[r]] : 0.0;
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) res[r] = 0.0;
    }
    double pap = 0.0;
    interior_block_core<R, FUSED, false, true>(w, xs, m, safe, word, slot0, oc, pap, res);
}

template <int R, bool FUSED, bool SRC>
__device__ __forceinline__ void interior_block(const double (&w)[15], const double *xs, int m, int safe, uint32_t word,
                                               int slot0, const double *sc, double *oc, double &pap, bool wdot = false)
{
    double acc[R];
    if (SRC) {
        int slot[R];
        block_slots<R>(m, word, slot0, slot);
        const int nv = (int)(word >> 28);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = r < nv ? sc[slot[r]] : 0.0;
        if (FUSED && wdot) {                 // the source values multiply (see class_items)
            double wsv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                wsv[r] = acc[r];
                acc[r] = 0.0;
            }
            interior_block_core<R, FUSED, true>(w, xs, m, safe, word, slot0, oc, pap, acc, wsv);
            return;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0;
    }
    interior_block_core<R, FUSED>(w, xs, m, safe, word, slot0, oc, pap, acc);
}

// One workgroup per coarse cell.  Load phase: coalesced column read, scatter into the LDS lattice
// image through the u16 position table.  Compute phase: surface slots (entity-major storage, class
// uniform per run) take their weight row from the LDS class table; the cell interior is swept in
// lattice order with SGPR-resident weights.  FUSED adds the CG smoother's neighbours to the same pass:
//   load phase   xin = x + beta*x2 (p = r + beta*p), written back to xout; block sum of xin*xin
//   epilogue     block sum of mult*xin*out  == this cell's share of dot(p, interface_sum(A p)), because
//                p is identical in all copies of a shared DOF (src/multigrid.jl:54-68)

}  // namespace hmg
