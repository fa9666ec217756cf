// How does the RELATIVE placement of concurrently streamed arrays move HBM throughput on MI355X?  (Round 3: the same binary runs a
// V-cycle in 130.4 or 137.7 ms depending on a 4 KB ... 4 MB shift between the level vectors.)  Three streams of one arena, c = a + b
// (double2 per thread, exact grid -- the launch shape of the CG vector kernels), a at 0, b at G + d, c at 2 G + 2 d (+ e for the
// second scan); GB/s over the offset d.
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
int main(int argc, char **argv)
{
    CK(hipSetDevice(0));
    const size_t G = (size_t)10294919168ull;         // a level-6 vector of config 3, rounded up to 2 MiB
    const size_t N = (size_t)8 << 30;                // bytes streamed per array
    char *arena;
    CK(hipMalloc((void **)&arena, 3 * G + ((size_t)4 << 30)));
    CK(hipMemset(arena, 0, 3 * G + ((size_t)4 << 30)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long n2 = (long)(N / 16);
    auto run = [&](size_t db, size_t dc) -> double {
        const double2 *a = (const double2 *)arena, *b = (const double2 *)(arena + G + db);
        double2 *c = (double2 *)(arena + 2 * G + dc);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_add, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, a, b, c, n2);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return 3.0 * N / (best * 1e-3) / 1e9;
    };
    printf("# b at G + d, c at 2 G + 2 d;   d [bytes]   GB/s\n");
    for (int s = 0; s <= 30; ++s) {
        const size_t d = s == 0 ? 0 : (size_t)1 << (s + 7);          // 0, 256 B ... 128 GiB/..: up to 2^37? capped below
        if (2 * d > ((size_t)3 << 30)) break;
        printf("d = 2^%-2d %12zu  %8.1f\n", s == 0 ? 0 : s + 7, d, run(d, 2 * d));
    }
    printf("# b at G + d, c at 2 G (only b moves)\n");
    for (int s = 0; s <= 30; ++s) {
        const size_t d = s == 0 ? 0 : (size_t)1 << (s + 7);
        if (d > ((size_t)3 << 30)) break;
        printf("d = 2^%-2d %12zu  %8.1f\n", s == 0 ? 0 : s + 7, d, run(d, 0));
    }
    printf("# b at G + k x 4 KiB, c at 2 G + 2 k x 4 KiB, k odd-ish\n");
    const size_t ks[] = {1, 3, 5, 9, 17, 33, 65, 129, 257, 513, 1025, 2049, 4097, 8193, 16385};
    for (size_t k : ks) printf("k = %6zu %12zu  %8.1f\n", k, k * 4096, run(k * 4096, 2 * k * 4096));
    return 0;
}
