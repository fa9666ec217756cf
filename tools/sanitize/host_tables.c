/*
 * host_tables.c -- every HOST code path behind the C ABI that runs before (and between) the first kernel launches, driven without a
 * GPU: mesh synthesis, the table builders of hmg_grid_create (NULL context: the uploads are checksummed instead of sent, see
 * DryUploads in csrc/hmg_capi.cpp), operator coefficients + cell classes, level-1 assembly, the domain shrink, and the partition
 * analysis (halo and global form) of every rank.  Built against the AddressSanitizer / UBSan / ThreadSanitizer builds of the
 * library by tests/test_sanitizers.py (`make -C homogenization.jl_amd/csrc asan tsan`), run with HMG_SETUP_THREADS = 1 / 3 / 16.
 *
 * Prints one "hash" line per grid: the checksum of all tables a device grid would have uploaded.  The same mesh must give the same
 * checksums whatever the thread count and whatever the allocator fills fresh memory with (ASAN_OPTIONS=malloc_fill_byte=..):
 * a checksum that moves is a data race or an uninitialised read.
 *
 *   host_tables [width = 12] [levels = 4] [ranks = 8]
 */
#include <stdio.h>
#include <stdlib.h>

#include "hmg.h"

#define CHECK(call)                                                            \
    do {                                                                       \
        if ((call) != 0) {                                                     \
            fprintf(stderr, "%s failed: %s\n", #call, hmg_last_error());       \
            return 1;                                                          \
        }                                                                      \
    } while (0)

static int print_hash(const char *what, hmg_grid *g)
{
    int32_t h[2] = {0, 0};
    int64_t n = 0;
    CHECK(hmg_grid_table_i32(g, 1, "upload_hash", h, 2, &n));
    printf("hash %-28s cells %8lld  %08x%08x\n", what, (long long)hmg_grid_ncells(g), (unsigned)h[1], (unsigned)h[0]);
    return 0;
}

int main(int argc, char **argv)
{
    const int w = argc > 1 ? atoi(argv[1]) : 12, levels = argc > 2 ? atoi(argv[2]) : 4, ranks = argc > 3 ? atoi(argv[3]) : 8;
    const int64_t shape[3] = {w, w, w};
    const double origin[3] = {-0.5 * w, -0.5 * w, -0.5 * w};
    int64_t nnodes = 0, ncells = 0;
    CHECK(hmg_checkerboard_mesh_size(3, shape, &nnodes, &ncells));
    double *coords = malloc(sizeof(double) * 3 * (size_t)nnodes);
    int64_t *cells = malloc(sizeof(int64_t) * 4 * (size_t)ncells);
    double *sigma = malloc(sizeof(double) * 3 * (size_t)ncells);
    double *sgrid = malloc(sizeof(double) * 3 * (size_t)w * w * w);
    int32_t *owner = malloc(sizeof(int32_t) * (size_t)ncells);
    CHECK(hmg_checkerboard_mesh(3, shape, origin, 1, 1, coords, cells));
    unsigned s = 12345u;
    for (int64_t q = 0; q < 3 * (int64_t)w * w * w; ++q) {
        s = s * 1664525u + 1013904223u;
        sgrid[q] = (s >> 16) & 1u ? 100.0 : 1.0;
    }
    const double off[3] = {0.5 * w + 1.0, 0.5 * w + 1.0, 0.5 * w + 1.0};
    CHECK(hmg_conductivity_per_element(3, nnodes, coords, ncells, cells, shape, sgrid, off, sigma));
    printf("host_tables: %d^3 cubes, %lld cells, %lld nodes, %d levels, %d ranks\n", w, (long long)ncells, (long long)nnodes, levels,
           ranks);

    /* the unpartitioned grid: tables, operator, level-1 matrix, shrink to the centred (w-2)^3 box */
    hmg_grid *g = NULL;
    CHECK(hmg_grid_create(NULL, 3, levels, nnodes, coords, ncells, cells, &g));
    if (print_hash("grid", g)) return 1;
    CHECK(hmg_grid_set_operator(g, sigma, 1.0));
    CHECK(hmg_coarse_setup(g));
    if (print_hash("grid + operator + level 1", g)) return 1;
    if (w > 2) {
        const int64_t v = w - 2, k = v + 1;
        CHECK(hmg_grid_shrink(g, 6 * v * v * v, k * k * k));
        CHECK(hmg_coarse_setup(g));
        if (print_hash("grid shrunk", g)) return 1;
    }
    CHECK(hmg_grid_destroy(g));

    /* every rank's share of the ownership partition (halves / quadrants / octants about the origin) */
    const int64_t blocks[3] = {ranks >= 2 ? 2 : 1, ranks >= 4 ? 2 : 1, ranks >= 8 ? 2 : 1};
    const int nr = (int)(blocks[0] * blocks[1] * blocks[2]);
    CHECK(hmg_block_owner(3, nnodes, coords, ncells, cells, blocks, 0.5 * w, origin, owner));
    for (int r = 0; r < nr; ++r) {
        hmg_grid *p = NULL;
        char name[64];
        CHECK(hmg_grid_create_partition(NULL, 3, levels, nnodes, coords, ncells, cells, owner, r, nr, &p));
        CHECK(hmg_grid_set_operator(p, sigma, 1.0));
        CHECK(hmg_coarse_setup(p));
        snprintf(name, sizeof name, "rank %d of %d", r, nr);
        if (print_hash(name, p)) return 1;
        if (w > 2 && r == nr - 1) {
            const int64_t v = w - 2, k = v + 1;
            CHECK(hmg_grid_shrink(p, 6 * v * v * v, k * k * k));
            if (print_hash("  ... shrunk", p)) return 1;
        }
        CHECK(hmg_grid_destroy(p));
    }
    /* the rehearsal form: one rank holds every cell, the cut is that of the octants */
    {
        hmg_grid *p = NULL;
        int32_t *zero = calloc((size_t)ncells, sizeof(int32_t));
        CHECK(hmg_grid_create_partition_rehearsal(NULL, 3, levels, nnodes, coords, ncells, cells, zero, owner, 0, 1, &p));
        CHECK(hmg_grid_set_operator(p, sigma, 1.0));
        if (print_hash("rehearsal", p)) return 1;
        CHECK(hmg_grid_destroy(p));
        free(zero);
    }
    free(owner);
    free(sgrid);
    free(sigma);
    free(cells);
    free(coords);
    printf("host_tables: done\n");
    return 0;
}
