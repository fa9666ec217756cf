// MFMA decision probe (BASELINE.json north_star: "MFMA only if recasting the per-cell block apply as a batched dense GEMM beats the
// LDS path in rocprof").  FP64 has one matrix instruction on gfx950, v_mfma_f64_16x16x4_f64.  Measured here, on the GPU box:
//   1. its issue rate (cycles per instruction on one SIMD, chip TFLOP/s) next to the v_fma_f64 rate of the vector pipe;
//   2. the cell interior of a level-6 cell (4495 nodes, 15-point stencil, one weight row per cell) as a banded GEMM from an LDS
//      lattice image: output tile = 16 consecutive i x 16 rows, D += A(16 x 4 Toeplitz block) * B(4 x 16 block of a source row),
//      7 source-row groups ((dj,dk) = (0,0) three taps, six groups of two taps) x 5 K-steps = 35 MFMAs per 256 outputs, 18 tiles
//      per cell if the tetrahedral lattice packed perfectly -- 3 x 512-thread workgroups per CU as in k_apply<3,512,13,*,6>.
// hipcc -O3 --offload-arch=gfx950 tools/dev/mfma_probe.hip -o tools/dev/mfma_probe && tools/dev/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_mfma_rate(double *out, int nit, long long *cyc)
{
    d4 c[8];
    for (int q = 0; q < 8; ++q) c[q] = d4{0, 0, 0, 0};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 0.5 + threadIdx.x * 1e-9;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < nit; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[q], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int q = 0; q < 8; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_fma_rate(double *out, int nit, long long *cyc)
{
    double c[8];
    for (int q = 0; q < 8; ++q) c[q] = q;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 0.5 + threadIdx.x * 1e-9;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < nit; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = __builtin_fma(a, c[q], b);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int q = 0; q < 8; ++q) s += c[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// banded-GEMM interior of one cell image per workgroup; nrep cells per workgroup to amortise the image fill
constexpr int NF = 6545, NTILE = 18;
__global__ void __launch_bounds__(512, 6) k_banded(const double *__restrict__ w15, double *out, int nrep)
{
    extern __shared__ double xs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = tid; q < NF + 128; q += 512) xs[q] = 1.0 + 1e-3 * ((q * 2654435761u) >> 20);
    // A fragments: group g (7 source rows), K-step s (inputs i0 - 2 + 4 s + kk): A[io][kk] = w_g[(4 s + kk - 2) - io + 1] inside the
    // band, formed on the fly from the cell's weights (uniform -> scalar registers): a few selects, hidden behind the MFMA
    double wu[15];
#pragma unroll
    for (int q = 0; q < 15; ++q) wu[q] = w15[q];
    const int io = lane & 15, kk = lane >> 4, d0 = kk - io - 1;
    __syncthreads();
    double acc_all = 0.0;
    for (int rep = 0; rep < nrep; ++rep) {
        for (int t = wave; t < NTILE; t += 8) {
            // 16 output rows of this tile (B column = lane & 15): lattice rows ~33 words apart; 16 consecutive i from i0
            const int row = ((t * 16 + (lane & 15)) * 33 + rep) % (NF - 120) + 40;
            d4 c = d4{0, 0, 0, 0};
#pragma unroll
            for (int g = 0; g < 7; ++g) {
                const int src = row + (g == 0 ? 0 : g & 1 ? 33 * ((g + 1) >> 1) : -33 * (g >> 1));   // a neighbouring row / plane
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    const double b = xs[src + 4 * s + kk - 2];
                    const int d = d0 + 4 * s;
                    const double a = d == 0 ? wu[2 * g] : d == 1 ? wu[2 * g + 1] : (d == 2 && g == 0) ? wu[14] : 0.0;
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
                }
            }
            acc_all += c[0] + c[1] + c[2] + c[3];
        }
    }
    out[(size_t)blockIdx.x * 512 + tid] = acc_all;
}

int main()
{
    CK(hipSetDevice(0));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    double *out;
    long long *cyc;
    CK(hipMalloc(&out, sizeof(double) * 512 * ncu * 64));
    CK(hipMalloc(&cyc, sizeof(long long) * ncu * 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<long long> hc(ncu * 64);
    float ms;
    const int nit = 20000;
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 0)
                hipLaunchKernelGGL(k_mfma_rate, dim3(ncu), dim3(256), 0, 0, out, nit, cyc);
            else
                hipLaunchKernelGGL(k_fma_rate, dim3(ncu), dim3(256), 0, 0, out, nit, cyc);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hc.data(), cyc, sizeof(long long) * ncu, hipMemcpyDeviceToHost));
        double mean = 0;
        for (int q = 0; q < ncu; ++q) mean += hc[q];
        mean /= ncu;
        const double n = (double)nit * 8;
        if (which == 0)
            printf("v_mfma_f64_16x16x4_f64: one wave per SIMD, %d CUs: %.1f s_memtime ticks (100 MHz) per 8 MFMAs x %d -> %.3f us; "
                   "%.2f TFLOP/s chip (2048 flop each); at 2.4 GHz that is %.1f cycles per MFMA\n",
                   ncu, mean / nit, nit, ms * 1e3 / 1.0, n * 4 * ncu * 2048.0 / (ms * 1e-3) / 1e12, ms * 1e-3 / n * 2.4e9);
        else
            printf("v_fma_f64            : one wave per SIMD, %d CUs: %.3f us total; %.2f TFLOP/s chip (128 flop per wave instruction); "
                   "at 2.4 GHz that is %.1f cycles per FMA instruction\n",
                   ncu, ms * 1e3, n * 4 * ncu * 128.0 / (ms * 1e-3) / 1e12, ms * 1e-3 / n * 2.4e9);
    }
    double hw[15], *dw;
    for (int q = 0; q < 15; ++q) hw[q] = 0.1 * (q + 1);
    CK(hipMalloc(&dw, sizeof(hw)));
    CK(hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice));
    const size_t lds = sizeof(double) * (NF + 128);
    CK(hipFuncSetAttribute((const void *)k_banded, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nrep = 64, nwg = ncu * 3 * 4;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_banded, dim3(nwg), dim3(512), lds, 0, dw, out, nrep);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double cells = (double)nwg * nrep;
    printf("banded-GEMM interior : %d workgroups x %d cell passes, %.3f ms -> %.2f us per cell and CU (18 tiles x 35 MFMAs + 35 ds_read_b64 "
           "per lane and tile; no loads from HBM, no surface, no stores)\n", nwg, nrep, ms, ms * 1e3 / (cells / ncu));
    printf("k_apply<3,512,13,*,6> today: 4.3 ms / 196608 cells x 256 CUs = 5.6 us per cell and CU for the WHOLE 16 B/DOF launch (HBM load, surface, "
           "interior, stores); its interior phase alone: 3.8 us per 512-thread workgroup, three of them overlapped\n");
    return 0;
}
