// How does the RELATIVE placement of concurrently streamed level vectors move HBM throughput on MI355X?  (Round 3: the same binary
// runs a V-cycle in 130.4 or 137.7 ms depending on a 4 KB ... 4 MB shift between the level vectors of one arena.)
//   scan 1: three flat streams c = a + b (double2 per thread, exact grid -- the shape of the CG vector kernels)
//   scan 2: the access pattern of the fused operator apply: one 512-thread workgroup per 52 360-byte column, three columns read
//           (r, p, x), three written (p', x', Ap), three workgroups resident per CU (53 KB of LDS each), no arithmetic
// arrays of one arena at k x (V + S), V = a level-6 vector of config 3 rounded up to 2 MiB; GB/s over S.
// hipcc -O3 --offload-arch=gfx950 tools/dev/offset_probe.hip -o tools/dev/offset_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_add(const double2 *__restrict__ a, const double2 *__restrict__ b, double2 *__restrict__ c, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const double2 x = a[i], y = b[i];
        c[i] = double2{x.x + y.x, x.y + y.y};
    }
}
constexpr int NF = 6545;
__global__ void __launch_bounds__(512, 6) k_col(const double *__restrict__ r, const double *p, const double *x, double *po, double *xo,
                                               double *ap, double beta)
{
    extern __shared__ double img[];
    const long base = (long)blockIdx.x * NF;
    double rv[13], pv[13], xv[13];
#pragma unroll
    for (int q = 0; q < 13; ++q) {
        const int t = threadIdx.x + q * 512;
        if (t < NF) {
            rv[q] = r[base + t];
            pv[q] = p[base + t];
            xv[q] = x[base + t];
        }
    }
#pragma unroll
    for (int q = 0; q < 13; ++q) {
        const int t = threadIdx.x + q * 512;
        if (t < NF) {
            const double v = rv[q] + beta * pv[q];
            po[base + t] = v;
            xo[base + t] = xv[q] + beta * pv[q];
            img[t] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 13; ++q) {
        const int t = threadIdx.x + q * 512;
        if (t < NF) ap[base + t] = img[NF - 1 - t];
    }
}
int main(int argc, char **argv)
{
    CK(hipSetDevice(0));
    const size_t V = (size_t)10294919168ull;
    const size_t PAD = (size_t)1 << 30;
    char *arena;
    CK(hipMalloc((void **)&arena, 6 * (V + PAD)));
    CK(hipMemset(arena, 0, 6 * (V + PAD)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t N = (size_t)8 << 30;
    const long n2 = (long)(N / 16);
    auto flat = [&](size_t S) -> double {
        const double2 *a = (const double2 *)arena, *b = (const double2 *)(arena + V + S);
        double2 *c = (double2 *)(arena + 2 * (V + S));
        float best = 1e30f, ms;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_add, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, a, b, c, n2);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return 3.0 * N / (best * 1e-3) / 1e9;
    };
    const int ncol = 196608;
    const size_t lds = sizeof(double) * (NF + 104);
    CK(hipFuncSetAttribute((const void *)k_col, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto cols = [&](size_t S, bool inplace) -> double {
        auto at = [&](int k) { return (double *)(arena + (size_t)k * (V + S)); };
        float best = 1e30f, ms;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (inplace)   // as the fused CG step: p and x updated in place, r read, Ap written
                hipLaunchKernelGGL(k_col, dim3(ncol), dim3(512), lds, 0, at(2), at(3), at(0), at(3), at(0), at(4), 0.5);
            else
                hipLaunchKernelGGL(k_col, dim3(ncol), dim3(512), lds, 0, at(0), at(1), at(2), at(3), at(4), at(5), 0.5);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return 6.0 * ncol * NF * 8.0 / (best * 1e-3) / 1e9;
    };
    const size_t ss[] = {0, 4, 8, 16, 32, 36, 64, 68, 128, 256, 260, 512, 1024, 1028, 2048, 4096, 4100, 8192, 16384, 16388, 32768, 65536, 131072, 262144, 524288};
    printf("# S [KB]   flat c=a+b GB/s   column kernel, 6 arrays GB/s   column kernel, in place (x,b,r,p,Ap order) GB/s\n");
    for (size_t kb : ss) {
        const size_t S = kb << 10;
        if (S > PAD) break;
        printf("%8zu  %8.1f  %8.1f  %8.1f\n", kb, flat(S), cols(S, false), cols(S, true));
        fflush(stdout);
    }
    return 0;
}
