/*
 * capi_vcycle.c -- the C ABI of libhmg_hip.so used from plain C (no Python, no torch): what a Julia `ccall`
 * host does, spelled out.  Solves -div(a grad u) + u = 1 with zero Dirichlet data on an
 * n^3-cube mesh of 6 n^3 tetrahedra with a few V-cycles and prints the residual norm after each.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/capi_vcycle.c -o capi_vcycle \
 *       -Lhomogenization.jl_amd -lhmg_hip -Wl,-rpath,$PWD/homogenization.jl_amd
 *   ./capi_vcycle [n = 4] [levels = 4] [cycles = 5]
 */
#define _XOPEN_SOURCE 700 /* sigaction, sigaltstack under -std=c99 */
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "hmg.h"

/* The program locates its own crashes: every ABI call leaves a breadcrumb on stderr (unbuffered) when HMG_EXAMPLE_TRACE is set or
 * after a fatal signal, and SIGSEGV / SIGBUS / SIGFPE / SIGILL / SIGABRT print the faulting address, the last ABI call entered and
 * a backtrace of the faulting thread (module + offset: resolve with `llvm-symbolizer -e <module> <offset>`), on a stack of their own. */
static const char *volatile last_call = "(none yet)";
static int trace_calls = 0;

static void put(const char *s) { (void)!write(2, s, strlen(s)); }

static void fatal_signal(int sig, siginfo_t *info, void *uctx)
{
    char buf[96];
    void *frames[64];
    (void)uctx;
    snprintf(buf, sizeof buf, "\ncapi_vcycle: fatal signal %d, fault address %p\n", sig, info ? info->si_addr : (void *)0);
    put(buf);
    put("capi_vcycle: last ABI call entered: ");
    put(last_call);
    put("\ncapi_vcycle: backtrace of the faulting thread:\n");
    backtrace_symbols_fd(frames, backtrace(frames, 64), 2);
    signal(sig, SIG_DFL); /* die of the same signal: the exit status stays what it was */
    raise(sig);
}

static void install_fatal_handlers(void)
{
    static char altstack[1 << 16];
    stack_t ss;
    struct sigaction sa;
    const int sigs[] = {SIGSEGV, SIGBUS, SIGFPE, SIGILL, SIGABRT};
    void *warm[1];
    (void)backtrace(warm, 1); /* loads libgcc's unwinder now, not inside the handler */
    ss.ss_sp = altstack;
    ss.ss_size = sizeof altstack;
    ss.ss_flags = 0;
    sigaltstack(&ss, NULL);
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = fatal_signal;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    for (size_t i = 0; i < sizeof sigs / sizeof sigs[0]; ++i) sigaction(sigs[i], &sa, NULL);
    trace_calls = getenv("HMG_EXAMPLE_TRACE") != NULL;
}

#define CHECK(call)                                                            \
    do {                                                                       \
        last_call = #call;                                                     \
        if (trace_calls) {                                                     \
            put("-> " #call "\n");                                             \
        }                                                                      \
        if ((call) != 0) {                                                     \
            fprintf(stderr, "%s failed: %s\n", #call, hmg_last_error());       \
            return 1;                                                          \
        }                                                                      \
    } while (0)

static int cmp_i64(const void *a, const void *b)
{
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

int main(int argc, char **argv)
{
    setvbuf(stdout, NULL, _IOLBF, 0); /* into a pipe as well: what was printed before a crash must not be lost with the buffer */
    install_fatal_handlers();
    if (argc > 1 && strcmp(argv[1], "--crash-selftest") == 0) { /* tests/test_c_abi_example.py: the evidence path itself */
        last_call = "(crash self-test)";
        raise(SIGSEGV);
    }
    const int n = argc > 1 ? atoi(argv[1]) : 4, levels = argc > 2 ? atoi(argv[2]) : 4;
    const int cycles = argc > 3 ? atoi(argv[3]) : 5;
    printf("capi_vcycle: %d^3 unit cubes, %d levels, %d V-cycles\n", n, levels, cycles);
    const int k = n + 1;
    const int64_t nnodes = (int64_t)k * k * k, ncells = 6 * (int64_t)n * n * n;
    /* every unit cube split into 6 tetrahedra around the diagonal 0-6 (corner c = bit pattern of +1 per axis) */
    static const int tets[6][4] = {{0, 1, 2, 6}, {0, 1, 4, 6}, {1, 3, 2, 6}, {1, 3, 6, 7}, {1, 5, 4, 6}, {1, 5, 6, 7}};
    double *coords = malloc(sizeof(double) * 3 * nnodes);
    int64_t *cells = malloc(sizeof(int64_t) * 4 * ncells);
    double *sigma = malloc(sizeof(double) * 3 * ncells);
    int64_t q = 0, c = 0;
    for (int x = 0; x < k; ++x)
        for (int y = 0; y < k; ++y)
            for (int z = 0; z < k; ++z, ++q) {
                coords[3 * q] = x;
                coords[3 * q + 1] = y;
                coords[3 * q + 2] = z;
            }
    for (int x = 0; x < n; ++x)
        for (int y = 0; y < n; ++y)
            for (int z = 0; z < n; ++z) {
                /* checkerboard: conductivity 1 or 9 per unit cube and direction */
                const unsigned h = (unsigned)(x * 73856093u ^ y * 19349663u ^ z * 83492791u);
                for (int t = 0; t < 6; ++t, ++c) {
                    for (int v = 0; v < 4; ++v) {
                        const int b = tets[t][v];
                        cells[4 * c + v] = 1 + ((int64_t)(x + (b & 1)) * k + (y + ((b >> 1) & 1))) * k + (z + ((b >> 2) & 1));
                    }
                    qsort(cells + 4 * c, 4, sizeof(int64_t), cmp_i64);   /* ascending tuples, 1-based */
                    for (int a = 0; a < 3; ++a) sigma[3 * c + a] = ((h >> a) & 1u) ? 9.0 : 1.0;
                }
            }

    hmg_ctx *ctx = NULL;
    hmg_grid *grid = NULL;
    CHECK(hmg_ctx_create(0, NULL, &ctx));
    CHECK(hmg_grid_create(ctx, 3, levels, nnodes, coords, ncells, cells, &grid));
    CHECK(hmg_grid_set_operator(grid, sigma, 1.0));
    CHECK(hmg_coarse_setup(grid));

    /* the library's default V-cycle form keeps a sixth vector on the finest level: reserved here, explicitly, so that a device
     * without room for it is an error at setup (hmg_grid_reserve_spare(grid, 0) would keep the reference's five vectors) */
    CHECK(hmg_grid_reserve_spare(grid, 1));
    /* LevelState(x, b, r, p, Ap) of every level */
    hmg_vec **st = calloc((size_t)5 * levels, sizeof(hmg_vec *));
    for (int l = 0; l < levels; ++l)
        for (int v = 0; v < 5; ++v) CHECK(hmg_vec_create(grid, l + 1, &st[5 * l + v]));
    hmg_vec *x = st[5 * (levels - 1)], *b = st[5 * (levels - 1) + 1], *r = st[5 * (levels - 1) + 2];
    CHECK(hmg_vec_fill_random(x, 1234, 0));
    CHECK(hmg_interface_sum(grid, levels, x));
    CHECK(hmg_constraint(grid, levels, x));
    CHECK(hmg_local_rhs(grid, b));

    printf("cells %lld, fine DOFs per cell %lld, levels %d\n", (long long)hmg_grid_ncells(grid),
           (long long)hmg_grid_nf(grid, levels), levels);
    double first = 0.0, last = 0.0;
    for (int i = 0; i < cycles; ++i) {
        CHECK(hmg_vcycle(grid, levels, 3, 2, st));
        CHECK(hmg_vec_norm_unique(r, &last));
        if (i == 0) first = last;
        printf("cycle %d  |r| = %.6e  (coarse PCG iterations %d)\n", i + 1, last, hmg_coarse_last_iterations(grid));
    }
    const int ok = last < first && last == last;

    for (int i = 0; i < 5 * levels; ++i) CHECK(hmg_vec_destroy(st[i]));
    CHECK(hmg_grid_destroy(grid));
    CHECK(hmg_ctx_destroy(ctx));
    last_call = "(all handles destroyed; freeing host arrays)";
    free(st);
    free(sigma);
    free(cells);
    free(coords);
    printf(ok ? "residual decreased: ok\n" : "residual did NOT decrease\n");
    last_call = "(main returned: exit handlers / static destructors of the loaded libraries)";
    return ok ? 0 : 2;
}
