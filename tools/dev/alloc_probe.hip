// How long does device memory allocation take on MI355X?  (setup time of the driver: 71 GB of level vectors)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    CK(hipSetDevice(0));
    CK(hipFree(0));
    const size_t GB = 1ull << 30;
    for (int rep = 0; rep < 2; ++rep) {
        std::vector<void *> p(6);
        double t0 = now();
        for (auto &q : p) CK(hipMalloc(&q, 10 * GB));
        double t1 = now();
        for (auto &q : p) CK(hipMemsetAsync(q, 0, 10 * GB, 0));
        CK(hipDeviceSynchronize());
        double t2 = now();
        for (auto &q : p) CK(hipFree(q));
        double t3 = now();
        printf("rep %d: 6 x hipMalloc(10 GB) %.3f s, memset %.3f s, hipFree %.3f s\n", rep, t1 - t0, t2 - t1, t3 - t2);
    }
    {
        void *q;
        double t0 = now();
        CK(hipMalloc(&q, 60 * GB));
        double t1 = now();
        CK(hipFree(q));
        printf("hipMalloc(60 GB) %.3f s, hipFree %.3f s\n", t1 - t0, now() - t1);
    }
    {
        hipStream_t s;
        CK(hipStreamCreate(&s));
        hipMemPool_t pool;
        CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = ~0ull;
        CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        for (int rep = 0; rep < 2; ++rep) {
            std::vector<void *> p(6);
            double t0 = now();
            for (auto &q : p) CK(hipMallocAsync(&q, 10 * GB, s));
            CK(hipStreamSynchronize(s));
            double t1 = now();
            for (auto &q : p) CK(hipFreeAsync(q, s));
            CK(hipStreamSynchronize(s));
            printf("rep %d: 6 x hipMallocAsync(10 GB) %.3f s, hipFreeAsync %.3f s\n", rep, t1 - t0, now() - t1);
        }
    }
    return 0;
}
