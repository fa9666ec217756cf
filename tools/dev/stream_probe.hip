// Streaming-kernel probe for MI355X: which launch shape / access form reaches the HBM ceiling for the CG vector
// updates (2 reads + 1 write + reduction, 24 B/DOF) and a plain copy (16 B/DOF)?   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef double double2_ __attribute__((ext_vector_type(2)));

template <int NT, int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(NT) k_triad(const double2_ *__restrict__ r, const double2_ *__restrict__ q, double2_ *__restrict__ o,
                                              long n2, double a, double *part)
{
    const long stride = (long)gridDim.x * NT;
    double acc = 0.0;
    long i = (long)blockIdx.x * NT + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        double2_ rv[U], qv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rv[u] = NTL ? __builtin_nontemporal_load(r + i + u * stride) : r[i + u * stride];
            qv[u] = NTL ? __builtin_nontemporal_load(q + i + u * stride) : q[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rv[u] = rv[u] - a * qv[u];
            if (NTS) __builtin_nontemporal_store(rv[u], o + i + u * stride); else o[i + u * stride] = rv[u];
            acc += rv[u].x * rv[u].x + rv[u].y * rv[u].y;
        }
    }
    for (; i < n2; i += stride) {
        double2_ rv = r[i] - a * q[i];
        o[i] = rv;
        acc += rv.x * rv.x + rv.y * rv.y;
    }
    for (int o2 = 32; o2 > 0; o2 >>= 1) acc += __shfl_down(acc, o2, 64);
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double s = 0; for (int w = 0; w < NT / 64; ++w) s += red[w]; part[blockIdx.x] = s; }
}

// chunked: every block owns one contiguous chunk (blockIdx-major), threads stride inside it
template <int NT, int U, bool NTS>
__global__ void __launch_bounds__(NT) k_triad_chunk(const double2_ *__restrict__ r, const double2_ *__restrict__ q, double2_ *__restrict__ o,
                                                    long n2, double a, double *part)
{
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long b = (long)blockIdx.x * per, e = b + per < n2 ? b + per : n2;
    double acc = 0.0;
    long i = b + threadIdx.x;
    for (; i + (U - 1) * NT < e; i += U * NT) {
        double2_ rv[U], qv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { rv[u] = r[i + u * NT]; qv[u] = q[i + u * NT]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rv[u] = rv[u] - a * qv[u];
            if (NTS) __builtin_nontemporal_store(rv[u], o + i + u * NT); else o[i + u * NT] = rv[u];
            acc += rv[u].x * rv[u].x + rv[u].y * rv[u].y;
        }
    }
    for (; i < e; i += NT) { double2_ rv = r[i] - a * q[i]; o[i] = rv; acc += rv.x * rv.x + rv.y * rv.y; }
    for (int o2 = 32; o2 > 0; o2 >>= 1) acc += __shfl_down(acc, o2, 64);
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double s = 0; for (int w = 0; w < NT / 64; ++w) s += red[w]; part[blockIdx.x] = s; }
}

// exact grid: block b owns elements [b*NT*U, (b+1)*NT*U), thread strides NT inside; no loop
template <int NT, int U, bool NTS>
__global__ void __launch_bounds__(NT) k_triad_exact(const double2_ *__restrict__ r, const double2_ *__restrict__ q, double2_ *__restrict__ o,
                                                    long n2, double a, double *part)
{
    const long b = (long)blockIdx.x * (NT * U) + threadIdx.x;
    double acc = 0.0;
    double2_ rv[U], qv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = b + u * NT;
        if (i < n2) { rv[u] = r[i]; qv[u] = q[i]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = b + u * NT;
        if (i < n2) {
            rv[u] = rv[u] - a * qv[u];
            if (NTS) __builtin_nontemporal_store(rv[u], o + i); else o[i] = rv[u];
            acc += rv[u].x * rv[u].x + rv[u].y * rv[u].y;
        }
    }
    for (int o2 = 32; o2 > 0; o2 >>= 1) acc += __shfl_down(acc, o2, 64);
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double s = 0; for (int w = 0; w < NT / 64; ++w) s += red[w]; part[blockIdx.x] = s; }
}
template <int NT, int U>
__global__ void __launch_bounds__(NT) k_copy_exact(const double2_ *__restrict__ s, double2_ *__restrict__ d, long n2)
{
    const long b = (long)blockIdx.x * (NT * U) + threadIdx.x;
    double2_ v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (b + u * NT < n2) v[u] = s[b + u * NT];
#pragma unroll
    for (int u = 0; u < U; ++u) if (b + u * NT < n2) d[b + u * NT] = v[u];
}

template <int NT, int U>
__global__ void __launch_bounds__(NT) k_copy(const double2_ *__restrict__ s, double2_ *__restrict__ d, long n2)
{
    const long stride = (long)gridDim.x * NT;
    long i = (long)blockIdx.x * NT + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        double2_ v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) d[i + u * stride] = v[u];
    }
    for (; i < n2; i += stride) d[i] = s[i];
}

template <int NT>
__global__ void __launch_bounds__(NT) k_read(const double2_ *__restrict__ s, long n2, double *part)
{
    const long stride = (long)gridDim.x * NT;
    double acc = 0;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n2; i += stride) { double2_ v = s[i]; acc += v.x + v.y; }
    if (acc == 1.2345) part[0] = acc;
}
template <int NT>
__global__ void __launch_bounds__(NT) k_write(double2_ *__restrict__ d, long n2)
{
    const long stride = (long)gridDim.x * NT;
    double2_ v = {1.0, 2.0};
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n2; i += stride) d[i] = v;
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 6545L * 196608L;
    const long n2 = n / 2;
    double *r, *q, *o, *part;
    CK(hipMalloc(&r, n * 8)); CK(hipMalloc(&q, n * 8)); CK(hipMalloc(&o, n * 8)); CK(hipMalloc(&part, 64 << 20));   // 8M block partials (largest grid below: n2 / 256 = 2.5M blocks)
    CK(hipMemset(r, 0, n * 8)); CK(hipMemset(q, 0, n * 8)); CK(hipMemset(o, 0, n * 8));
    hipLaunchKernelGGL(k_write<256>, dim3(2048), dim3(256), 0, 0, (double2_ *)r, n2);
    hipLaunchKernelGGL(k_write<256>, dim3(2048), dim3(256), 0, 0, (double2_ *)q, n2);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cu = prop.multiProcessorCount;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double bytes, auto &&launch) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
        }
        CK(hipGetLastError());
        printf("%-58s best %7.3f ms  avg %7.3f ms  %6.0f GB/s (best)\n", name, best, sum / 5, bytes / best / 1e6);
        fflush(stdout);
    };
    const double a = 0.37;
    const double B3 = 24.0 * n, B2 = 16.0 * n;
    double2_ *r2 = (double2_ *)r, *q2 = (double2_ *)q, *o2 = (double2_ *)o;
#define TRIAD(NT, U, NTL, NTS, BLOCKS, label) timeit(label, B3, [&] { hipLaunchKernelGGL((k_triad<NT, U, NTL, NTS>), dim3(BLOCKS), dim3(NT), 0, 0, r2, q2, o2, n2, a, part); })
    TRIAD(256, 1, false, false, cu * 8,  "triad 256thr U1 grid 8/CU (current)");
    TRIAD(256, 2, false, false, cu * 8,  "triad 256thr U2 grid 8/CU");
    TRIAD(256, 4, false, false, cu * 8,  "triad 256thr U4 grid 8/CU");
    TRIAD(256, 4, false, false, cu * 4,  "triad 256thr U4 grid 4/CU");
    TRIAD(256, 4, false, false, cu * 16, "triad 256thr U4 grid 16/CU");
    TRIAD(512, 4, false, false, cu * 4,  "triad 512thr U4 grid 4/CU");
    TRIAD(1024, 4, false, false, cu * 2, "triad 1024thr U4 grid 2/CU");
    TRIAD(1024, 4, false, false, cu * 1, "triad 1024thr U4 grid 1/CU");
    TRIAD(1024, 2, false, false, cu * 2, "triad 1024thr U2 grid 2/CU");
    TRIAD(1024, 8, false, false, cu * 2, "triad 1024thr U8 grid 2/CU");
    TRIAD(256, 4, false, true,  cu * 8,  "triad 256thr U4 grid 8/CU nt-store");
    TRIAD(256, 4, true,  true,  cu * 8,  "triad 256thr U4 grid 8/CU nt-load nt-store");
    TRIAD(1024, 4, false, true, cu * 2,  "triad 1024thr U4 grid 2/CU nt-store");
    TRIAD(1024, 4, true, true,  cu * 2,  "triad 1024thr U4 grid 2/CU nt-load nt-store");
    TRIAD(256, 4, false, false, 65536,   "triad 256thr U4 grid 65536");
    TRIAD(256, 1, false, false, 1 << 20, "triad 256thr U1 grid 1M");
#define CHUNK(NT, U, NTS, BLOCKS, label) timeit(label, B3, [&] { hipLaunchKernelGGL((k_triad_chunk<NT, U, NTS>), dim3(BLOCKS), dim3(NT), 0, 0, r2, q2, o2, n2, a, part); })
    CHUNK(256, 4, false, cu * 8,  "triad chunked 256thr U4 grid 8/CU");
    CHUNK(1024, 4, false, cu * 2, "triad chunked 1024thr U4 grid 2/CU");
    CHUNK(256, 4, false, 65536,   "triad chunked 256thr U4 grid 65536");
    CHUNK(256, 4, true, 65536,    "triad chunked 256thr U4 grid 65536 nt-store");
    CHUNK(512, 4, false, 196608,  "triad chunked 512thr U4 grid 196608 (one per cell)");
#define EXACT(NT, U, NTS, label) timeit(label, B3, [&] { hipLaunchKernelGGL((k_triad_exact<NT, U, NTS>), dim3((unsigned)((n2 + NT * U - 1) / (NT * U))), dim3(NT), 0, 0, r2, q2, o2, n2, a, part); })
    EXACT(256, 1, false, "triad exact 256thr U1");
    EXACT(256, 2, false, "triad exact 256thr U2");
    EXACT(256, 4, false, "triad exact 256thr U4");
    EXACT(256, 8, false, "triad exact 256thr U8");
    EXACT(512, 2, false, "triad exact 512thr U2");
    EXACT(512, 4, false, "triad exact 512thr U4");
    EXACT(1024, 2, false, "triad exact 1024thr U2");
    EXACT(1024, 4, false, "triad exact 1024thr U4");
    EXACT(128, 4, false, "triad exact 128thr U4");
    EXACT(64, 4, false, "triad exact 64thr U4");
    EXACT(256, 4, true, "triad exact 256thr U4 nt-store");
    timeit("triad exact in place 256thr U4", B3, [&] { hipLaunchKernelGGL((k_triad_exact<256, 4, false>), dim3((unsigned)((n2 + 1023) / 1024)), dim3(256), 0, 0, r2, q2, r2, n2, a, part); });
    timeit("copy exact 256thr U4", B2, [&] { hipLaunchKernelGGL((k_copy_exact<256, 4>), dim3((unsigned)((n2 + 1023) / 1024)), dim3(256), 0, 0, r2, o2, n2); });
    timeit("copy exact 256thr U1", B2, [&] { hipLaunchKernelGGL((k_copy_exact<256, 1>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, r2, o2, n2); });
    timeit("copy exact 512thr U4", B2, [&] { hipLaunchKernelGGL((k_copy_exact<512, 4>), dim3((unsigned)((n2 + 2047) / 2048)), dim3(512), 0, 0, r2, o2, n2); });
    // in place: o == r
    timeit("triad in place (o = r) 256thr U4 grid 8/CU", B3, [&] { hipLaunchKernelGGL((k_triad<256, 4, false, false>), dim3(cu * 8), dim3(256), 0, 0, r2, q2, r2, n2, a, part); });
    timeit("copy 256thr U1 grid 8/CU", B2, [&] { hipLaunchKernelGGL((k_copy<256, 1>), dim3(cu * 8), dim3(256), 0, 0, r2, o2, n2); });
    timeit("copy 256thr U4 grid 8/CU", B2, [&] { hipLaunchKernelGGL((k_copy<256, 4>), dim3(cu * 8), dim3(256), 0, 0, r2, o2, n2); });
    timeit("copy 1024thr U4 grid 2/CU", B2, [&] { hipLaunchKernelGGL((k_copy<1024, 4>), dim3(cu * 2), dim3(1024), 0, 0, r2, o2, n2); });
    timeit("copy 256thr U4 grid 65536", B2, [&] { hipLaunchKernelGGL((k_copy<256, 4>), dim3(65536), dim3(256), 0, 0, r2, o2, n2); });
    timeit("hipMemcpyAsync D2D", B2, [&] { CK(hipMemcpyAsync(o, r, n * 8, hipMemcpyDeviceToDevice, 0)); });
    timeit("read only 256thr grid 8/CU", 8.0 * n, [&] { hipLaunchKernelGGL((k_read<256>), dim3(cu * 8), dim3(256), 0, 0, r2, n2, part); });
    timeit("write only 256thr grid 8/CU", 8.0 * n, [&] { hipLaunchKernelGGL((k_write<256>), dim3(cu * 8), dim3(256), 0, 0, o2, n2); });
    timeit("hipMemsetAsync", 8.0 * n, [&] { CK(hipMemsetAsync(o, 0, n * 8, 0)); });
    return 0;
}
