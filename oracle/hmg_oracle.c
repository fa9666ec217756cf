/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the hot loops of haampie/Homogenization.jl (the level-vector
 * kernels of the matrix-free multigrid on the implicit fine grid). Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library;
 * the product path (homogenization.jl_amd/) never does.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference checkout). Indices are 0-based here (the reference is 1-based); loop orders and
 * floating-point operation orders follow the reference statement by statement.
 *
 * Parity pin: see oracle/oracle.py header.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t i64;

/* y[:, col] += alpha * A * x[:, col], CSC column scatter.
 * ref: src/apply_local_operators.jl:125-133 (my_A_mul_B!) */
static inline void csc_scatter(double alpha, i64 n, const i64 *colptr, const i64 *rowval,
                               const double *nzval, const double *x, double *y, i64 offset)
{
    for (i64 j = 0; j < n; ++j) {
        double axj = alpha * x[j + offset];
        for (i64 i = colptr[j]; i < colptr[j + 1]; ++i)
            y[rowval[i] + offset] += nzval[i] * axj;
    }
}

void orc_csc_scatter(double alpha, i64 n, const i64 *colptr, const i64 *rowval,
                     const double *nzval, const double *x, double *y, i64 offset)
{
    csc_scatter(alpha, n, colptr, rowval, nzval, x, y, offset);
}

/* y += alpha * A * x for A = lambda*M - div(sigma grad), cell by cell.
 * ref: src/apply_local_operators.jl:85-120 (mul! + do_share_of_mv_product! for L2PlusDivAGrad).
 * Threading: static cyclic distribution of cells over nthreads, as :88-98.
 *
 * jinv   : per cell dim*dim, column-major, = inv(J')  (ref: src/cell_values.jl:113)
 * detj   : per cell |det J|                           (ref: src/cell_values.jl:121)
 * sigma  : per cell dim entries
 * ops_*  : dim*dim CSC matrices, matrix (i,j) (0-based) at slot i + dim*j, each nf x nf.
 *          All colptr arrays are concatenated with stride (nf+1); rowval/nzval via ops_base.
 */
void orc_apply_l2divagrad(double alpha, int dim, i64 ncells, i64 nf,
                          const double *jinv, const double *detj, const double *sigma,
                          double lambda,
                          const i64 *ops_colptr, const i64 *ops_base,
                          const i64 *ops_rowval, const double *ops_nzval,
                          const i64 *m_colptr, const i64 *m_rowval, const double *m_nzval,
                          const double *x, double *y, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int t = 0; t < nthreads; ++t) {
        for (i64 el = t; el < ncells; el += nthreads) {
            const double *Ji = jinv + el * dim * dim; /* column-major dim x dim */
            const double *sg = sigma + el * dim;
            double det = detj[el];
            double P[9];
            /* P = Jinv' * (sigma .* Jinv)  (:105); sigma scales the rows of Jinv. */
            for (int i = 0; i < dim; ++i)
                for (int j = 0; j < dim; ++j) {
                    double s = 0.0;
                    for (int k = 0; k < dim; ++k)
                        s += Ji[k + dim * i] * (sg[k] * Ji[k + dim * j]);
                    P[i + dim * j] = s;
                }
            i64 offset = el * nf; /* :108 */
            for (int i = 0; i < dim; ++i)       /* :111 for i = 1:dim, j = 1:dim */
                for (int j = 0; j < dim; ++j) {
                    int slot = i + dim * j;
                    csc_scatter(alpha * det * P[i + dim * j], nf, ops_colptr + slot * (nf + 1),
                                ops_rowval + ops_base[slot], ops_nzval + ops_base[slot], x, y,
                                offset);
                }
            double am = alpha * lambda * det;   /* :116 */
            if (am != 0.0) csc_scatter(am, nf, m_colptr, m_rowval, m_nzval, x, y, offset);
        }
    }
}

/* SimpleDiffusion twin: coefficient alpha * P[i,j] * detJ * a with P = Jinv' * Jinv.
 * ref: src/apply_local_operators.jl:40-72 */
void orc_apply_simple_diffusion(double alpha, int dim, i64 ncells, i64 nf, const double *jinv,
                                const double *detj, double a, const i64 *ops_colptr,
                                const i64 *ops_base, const i64 *ops_rowval,
                                const double *ops_nzval, const double *x, double *y,
                                int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int t = 0; t < nthreads; ++t) {
        for (i64 el = t; el < ncells; el += nthreads) {
            const double *Ji = jinv + el * dim * dim;
            double det = detj[el];
            double P[9];
            for (int i = 0; i < dim; ++i)
                for (int j = 0; j < dim; ++j) {
                    double s = 0.0;
                    for (int k = 0; k < dim; ++k) s += Ji[k + dim * i] * Ji[k + dim * j];
                    P[i + dim * j] = s;
                }
            i64 offset = el * nf;
            for (int i = 0; i < dim; ++i)
                for (int j = 0; j < dim; ++j) {
                    int slot = i + dim * j;
                    csc_scatter(alpha * P[i + dim * j] * det * a, nf,
                                ops_colptr + slot * (nf + 1), ops_rowval + ops_base[slot],
                                ops_nzval + ops_base[slot], x, y, offset);
                }
        }
    }
}

/* Sparse cell->element map in CSR form (ref: src/interface.jl:31-47):
 *   off[ncell_entities+1], val_el[], val_lid[]   (0-based element, 0-based local id)
 * Local numbering lists in CSR form: num_ptr[nlocal+1], num_idx[] (0-based node ids).
 */

/* Sum-and-replicate over one entity class (faces, edges or nodes).
 * ref: src/implicit_fine_grid.jl:219-251 (faces), :260-292 (edges), :297-325 (nodes) */
void orc_broadcast_class(double *x, i64 nf, i64 nent, const i64 *off, const i64 *val_el,
                         const i64 *val_lid, const i64 *num_ptr, const i64 *num_idx,
                         double *buffer)
{
    for (i64 i = 0; i < nent; ++i) {
        i64 first = off[i];
        if (off[i + 1] == first) continue;
        i64 l0 = val_lid[first];
        i64 per = num_ptr[l0 + 1] - num_ptr[l0];
        for (i64 k = 0; k < per; ++k) buffer[k] = 0.0;
        for (i64 j = off[i]; j < off[i + 1]; ++j) { /* Reduce */
            const i64 *nodes = num_idx + num_ptr[val_lid[j]];
            i64 base = val_el[j] * nf;
            for (i64 k = 0; k < per; ++k) buffer[k] += x[nodes[k] + base];
        }
        for (i64 j = off[i]; j < off[i + 1]; ++j) { /* Broadcast */
            const i64 *nodes = num_idx + num_ptr[val_lid[j]];
            i64 base = val_el[j] * nf;
            for (i64 k = 0; k < per; ++k) x[nodes[k] + base] = buffer[k];
        }
    }
}

/* Zero the listed (element, local entity) DOFs.
 * all_copies = 1: every listed copy      -> apply_constraint! (src/implicit_fine_grid.jl:94-139)
 * all_copies = 0: copies 2..n per entity -> zero_out_all_but_one! (:334-386) */
void orc_zero_class(double *x, i64 nf, i64 nent, const i64 *off, const i64 *val_el,
                    const i64 *val_lid, const i64 *num_ptr, const i64 *num_idx, int all_copies)
{
    for (i64 i = 0; i < nent; ++i) {
        i64 j0 = all_copies ? off[i] : off[i] + 1;
        for (i64 j = j0; j < off[i + 1]; ++j) {
            i64 l = val_lid[j];
            i64 base = val_el[j] * nf;
            for (i64 k = num_ptr[l]; k < num_ptr[l + 1]; ++k) x[num_idx[k] + base] = 0.0;
        }
    }
}

/* u[global node] = v[local node, first listed element]; ref: src/implicit_fine_grid.jl:148-171 */
void orc_copy_to_base(double *u, const double *v, i64 nf, i64 nent, const i64 *cells,
                      const i64 *off, const i64 *val_el, const i64 *val_lid, const i64 *numbering_nodes)
{
    for (i64 i = 0; i < nent; ++i) {
        i64 j = off[i];
        u[cells[i]] = v[numbering_nodes[val_lid[j]] + val_el[j] * nf];
    }
}

/* v[local node, element] = u[global node] for all copies; ref: src/implicit_fine_grid.jl:178-202 */
void orc_distribute(double *v, const double *u, i64 nf, i64 nent, const i64 *cells,
                    const i64 *off, const i64 *val_el, const i64 *val_lid, const i64 *numbering_nodes)
{
    for (i64 i = 0; i < nent; ++i) {
        double uv = u[cells[i]];
        for (i64 j = off[i]; j < off[i + 1]; ++j)
            v[numbering_nodes[val_lid[j]] + val_el[j] * nf] = uv;
    }
}

/* y[:,col] += P * x[:,col] for every column (P is CSC, nfine x ncoarse).
 * ref: src/interpolation.jl:64-74 -> 5-arg mul!(y, P, x, 1, 1): CSC column scatter. */
void orc_interpolate_and_sum(double *y, i64 nfine, const double *x, i64 ncoarse, i64 ncols,
                             const i64 *colptr, const i64 *rowval, const double *nzval,
                             int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int t = 0; t < nthreads; ++t)
        for (i64 col = t; col < ncols; col += nthreads) {
            double *yc = y + col * nfine;
            const double *xc = x + col * ncoarse;
            for (i64 j = 0; j < ncoarse; ++j) {
                double axj = xc[j] * 1.0;
                for (i64 i = colptr[j]; i < colptr[j + 1]; ++i) yc[rowval[i]] += nzval[i] * axj;
            }
        }
}

/* y[:,col] = P' * x[:,col] (overwrite). ref: src/interpolation.jl:52-62 -> mul!(y, P', x):
 * per column of P a dot product accumulated in ascending row order. */
void orc_restrict(double *y, i64 ncoarse, const double *x, i64 nfine, i64 ncols,
                  const i64 *colptr, const i64 *rowval, const double *nzval, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int t = 0; t < nthreads; ++t)
        for (i64 col = t; col < ncols; col += nthreads) {
            double *yc = y + col * ncoarse;
            const double *xc = x + col * nfine;
            for (i64 j = 0; j < ncoarse; ++j) {
                double tmp = 0.0;
                for (i64 i = colptr[j]; i < colptr[j + 1]; ++i) tmp += nzval[i] * xc[rowval[i]];
                yc[j] = tmp;
            }
        }
}

/* BLAS-1 stand-ins for the OpenBLAS calls of src/multigrid.jl:54,64-68 (dot, axpy!) and the
 * broadcast p .= r .+ c .* p (:68). Plain loops over the raw storage (shared DOFs counted once
 * per copy, exactly as BLAS on the Nf x Ne matrix does). */
double orc_dot(i64 n, const double *x, const double *y, int nthreads)
{
    double s = 0.0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) reduction(+ : s) schedule(static)
    for (i64 i = 0; i < n; ++i) s += x[i] * y[i];
    return s;
}

void orc_axpy(i64 n, double a, const double *x, double *y, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (i64 i = 0; i < n; ++i) y[i] += a * x[i];
}

void orc_xpby(i64 n, const double *r, double c, double *p, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (i64 i = 0; i < n; ++i) p[i] = r[i] + c * p[i];
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
