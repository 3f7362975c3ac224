// Correctness + timing probe of csdh_kernel (K4 on the half-precision matrix cores, csrc/csdh_kernel.h) against a float64
// product and against the float32 3-multiplication kernel (development aid, torch-free).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/csdh_probe tools/csdh_probe.hip
//   tools/bin/csdh_probe [rows=7000] [F=512] [mode=0]      mode 0: AR-like noise, channel gains 1e-14 ... 1e6
//                                                         mode 1: + 80 dB spectral tilt (validity fallback expected off)
//                                                         mode 2: + one channel with a 130 dB line (fallback expected)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define SPY_DYN_SMEM(type, name) extern __shared__ __attribute__((aligned(16))) char name##_raw[]; \
    type* name = reinterpret_cast<type*>(name##_raw)
#include "../include/spyhip.h"
#include "../syncopy_amd/csrc/csd_kernel.h"
#include "../syncopy_amd/csrc/csdh_kernel.h"
#include "csdhp_kernel.h"

__device__ __forceinline__ float hash_unit(unsigned long long i) {
    unsigned long long h = i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
    return (float)(unsigned)(h & 0xffffffu) * (1.f / 8388608.f) - 1.f;      // [-1, 1)
}

// X[r, f, c] = gain[c] * tilt[f] * (noise + coupling to a common source of the row)
__global__ void fill(float2* p, long long rows, int F, int mode) {
    const long long n = rows * F * 256;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i & 255);
        const long long rf = i >> 8;
        const int f = (int)(rf % F);
        float g = exp2f((float)((c * 37) % 67) - 46.f);                       // 2^-46 ... 2^20
        float tilt = 1.f;
        if (mode >= 1) tilt = exp2f(-13.3f * (float)f / (float)F);            // 80 dB in amplitude
        if (mode >= 2 && c == 5) tilt = (f == 3) ? 1.f : exp2f(-21.6f);        // a 130 dB line on channel 5
        const float sr = hash_unit(2 * rf), si = hash_unit(2 * rf + 1);       // common source
        const float nr = hash_unit(0x100000000ull + 2 * i) + hash_unit(0x300000000ull + 2 * i);
        const float ni = hash_unit(0x100000000ull + 2 * i + 1) + hash_unit(0x300000000ull + 2 * i + 1);
        const float k = (c % 3 == 0) ? 0.9f : 0.05f;
        p[i] = make_float2(g * tilt * (nr + k * sr), g * tilt * (ni + k * si));
    }
}

// float64 product of one frequency: ref[i, j] = sum_r X[r, f, i] conj(X[r, f, j])
__global__ void ref64(const float2* spec, long long rows, int F, int f, double2* out) {
    const int i = blockIdx.x, j = threadIdx.x;
    double re = 0.0, im = 0.0;
    for (long long r = 0; r < rows; ++r) {
        const float2 a = spec[(r * F + f) * 256 + i], b = spec[(r * F + f) * 256 + j];
        re += (double)a.x * b.x + (double)a.y * b.y;
        im += (double)a.y * b.x - (double)a.x * b.y;
    }
    out[i * 256 + j] = make_double2(re, im);
}

int main(int argc, char** argv) {
    const long long rows = argc > 1 ? atoll(argv[1]) : 7000;
    const int F = argc > 2 ? atoi(argv[2]) : 512, mode = argc > 3 ? atoi(argv[3]) : 0;
    const int reps = argc > 4 ? atoi(argv[4]) : 3;
    float2 *spec, *acc, *acc3;
    float* absmax;
    int* flags;
    double2* ref;
    hipMalloc(&spec, (size_t)rows * F * 256 * 8);
    hipMalloc(&acc, (size_t)F * 65536 * 8);
    hipMalloc(&acc3, (size_t)F * 65536 * 8);
    hipMalloc(&absmax, 1024);
    hipMalloc(&flags, F * 4);
    hipMalloc(&ref, 65536 * 16);
    fill<<<8192, 256>>>(spec, rows, F, mode);
    hipMemset(acc, 0, (size_t)F * 65536 * 8);
    hipMemset(acc3, 0, (size_t)F * 65536 * 8);
    hipMemset(absmax, 0, 1024);
    hipMemset(flags, 0xff, F * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;

    const long long nquads = rows * F * 128;
    hipEventRecord(e0);
    spycsd::csdh_absmax_kernel<<<2048, 256>>>((const float4*)spec, nquads, 256, (unsigned*)absmax);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("absmax pass: %.3f ms (%.2f TB/s) %s\n", ms, (double)nquads * 16 / ms * 1e-9, hipGetErrorString(hipGetLastError()));

    spycsd::CsdhArgs a{};
    a.spec = spec; a.nrows = rows; a.F = F; a.acc = acc; a.absmax = absmax; a.flags = flags; a.f0 = 0; a.nf = F; a.rs = (long long)F * 256; a.fs = 256;
    if (getenv("CSDH_FMAJOR")) { a.rs = 256; a.fs = rows * 256; }
    if (getenv("CSDH_BLOCKED")) { a.rs = 256; a.fs = 32 * 256; /* rows of a 32-row block contiguous; blocks overlap: timing only */ }
    unsigned long long* stamps;
    hipMalloc(&stamps, 4 * 8 * 8 * 8);
    hipMemset(stamps, 0, 4 * 8 * 8 * 8);
    a.stamps = stamps;
    hipFuncSetAttribute((const void*)spycsd::csdh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, spycsd::CSDH_LDS_BYTES);
    spycsd::csdh_kernel<<<F, 512, spycsd::CSDH_LDS_BYTES>>>(a);
    hipDeviceSynchronize();
    printf("csdh first launch: %s\n", hipGetErrorString(hipGetLastError()));

#ifdef CSDH_STAMPS
    {
        std::vector<unsigned long long> h(4 * 8 * 8);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        const unsigned long long t0 = h[0];
        printf("timeline of block 0 (cycles since wave 0 entered chunk 20): points 0 chunk start, 1 after first tile, 2 before / 3 after the wait for loads A, 4 after conversion A + issue of loads B, 5 before / 6 after the wait for loads B, 7 before the barrier\n");
        for (int c = 0; c < 4; ++c)
            for (int g = 0; g < 8; ++g) {
                printf("chunk %d wave %d:", 20 + c, g);
                for (int k = 0; k < 8; ++k) printf(" %7lld", (long long)(h[(c * 8 + g) * 8 + k] - t0));
                printf("\n");
            }
    }
#endif
    // the same through the pre-split hand-over (csdhp_kernel.h): planes from csdhp_split_kernel, LDS-DMA, transpose reads
    float2* accp;
    uint4* planes;
    hipMalloc(&accp, (size_t)F * 65536 * 8);
    hipMemset(accp, 0, (size_t)F * 65536 * 8);
    hipMalloc(&planes, (size_t)rows * F * 2048);
    spycsd::csdhp_split_kernel<<<8192, 256>>>((const float4*)spec, rows * F * 64, absmax, planes);
    spycsd::CsdhArgs ap = a;
    ap.spec = (const float2*)planes; ap.acc = accp; ap.rs = (long long)F * 256; ap.fs = 256;
    hipFuncSetAttribute((const void*)spycsd::csdhp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, spycsd::CSDHP_LDS_BYTES);
    spycsd::csdhp_kernel<<<F, 512, spycsd::CSDHP_LDS_BYTES>>>(ap);
    hipDeviceSynchronize();
    printf("csdhp first launch: %s\n", hipGetErrorString(hipGetLastError()));
    {
        std::vector<float2> h0(65536), h1(65536);
        double worst = 0;
        for (int f : {0, 1, F / 2, F - 1}) {
            hipMemcpy(h0.data(), acc + (size_t)f * 65536, 65536 * 8, hipMemcpyDeviceToHost);
            hipMemcpy(h1.data(), accp + (size_t)f * 65536, 65536 * 8, hipMemcpyDeviceToHost);
            for (int i = 0; i < 256; ++i)
                for (int j = 0; j <= i; ++j) {
                    const double n = std::sqrt((double)h0[i * 256 + i].x * h0[j * 256 + j].x);
                    if (n == 0) continue;
                    worst = std::max(worst, std::hypot((double)h1[i * 256 + j].x - h0[i * 256 + j].x, (double)h1[i * 256 + j].y - h0[i * 256 + j].y) / n);
                }
        }
        printf("pre-split hand-over vs conversion in the kernel: max |diff| / sqrt(Sii Sjj) = %.3e\n", worst);
    }
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) spycsd::csdhp_kernel<<<F, 512, spycsd::CSDHP_LDS_BYTES>>>(ap);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("csdhp: rows=%lld F=%d: %.3f ms, %.1f TF algorithmic, %.1f TF fp16 executed   %s\n", rows, F, ms,
           8.0 * rows * F * 256 * 257 / 2 / ms / 1e9, (double)((rows + 31) / 32) * F * 136 * 12 * 16384.0 / ms / 1e9,
           hipGetErrorString(hipGetLastError()));

    // float32 3M kernel for comparison
    spycsd::CsdArgs b{};
    b.spec = spec; b.nrows = rows; b.F = F; b.C = 256; b.acc = acc3;
    b.nt = 8; b.ntiles = 36; b.nitems = (long long)F * 36; b.cpad = 256; b.item_base = 0; b.item_end = b.nitems;
    auto k3 = spycsd::csd3m_kernel<256, 8>;
    hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, spycsd::M3_LDS_BYTES);
    k3<<<F, 512, spycsd::M3_LDS_BYTES>>>(b);
    hipDeviceSynchronize();

    std::vector<int> hflags(F);
    hipMemcpy(hflags.data(), flags, F * 4, hipMemcpyDeviceToHost);
    int nflag = 0;
    for (int f = 0; f < F; ++f) nflag += hflags[f] != 0;
    printf("flagged (not committed) frequencies: %d of %d\n", nflag, F);

    // accuracy at a few frequencies
    std::vector<double2> href(65536);
    std::vector<float2> hacc(65536), hacc3(65536);
    const int fs[6] = {0, 1, 3, F / 2, F - 2, F - 1};
    for (int q = 0; q < 6; ++q) {
        const int f = fs[q];
        if (f < 0 || f >= F) continue;
        ref64<<<256, 256>>>(spec, rows, F, f, ref);
        hipMemcpy(href.data(), ref, 65536 * 16, hipMemcpyDeviceToHost);
        hipMemcpy(hacc.data(), acc + (size_t)f * 65536, 65536 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hacc3.data(), acc3 + (size_t)f * 65536, 65536 * 8, hipMemcpyDeviceToHost);
        double eh = 0, e3 = 0, ehi = 0, e3i = 0;
        int wi = 0, wj = 0;
        for (int i = 0; i < 256; ++i)
            for (int j = 0; j <= i; ++j) {
                const double nrm = std::sqrt(href[i * 256 + i].x * href[j * 256 + j].x);
                if (nrm == 0) continue;
                const double dr = hacc[i * 256 + j].x - href[i * 256 + j].x, di = (i == j) ? 0.0 : hacc[i * 256 + j].y - href[i * 256 + j].y;
                const double d3r = hacc3[i * 256 + j].x - href[i * 256 + j].x, d3i = (i == j) ? 0.0 : hacc3[i * 256 + j].y - href[i * 256 + j].y;
                const double e = std::sqrt(dr * dr + di * di) / nrm;
                if (e > eh) { eh = e; wi = i; wj = j; }
                e3 = std::max(e3, std::sqrt(d3r * d3r + d3i * d3i) / nrm);
                ehi = std::max(ehi, std::fabs(di) / nrm);
                e3i = std::max(e3i, std::fabs(d3i) / nrm);
            }
        printf("f=%4d flag=%d  max |S - S64| / sqrt(Sii Sjj): half-split %.3e (imag %.3e) at (%d,%d)   float32 3M %.3e (imag %.3e)\n",
               f, hflags[f], eh, ehi, wi, wj, e3, e3i);
    }

    // timing
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) spycsd::csdh_kernel<<<F, 512, spycsd::CSDH_LDS_BYTES>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double fl = 8.0 * rows * F * 256 * 257 / 2;
    printf("csdh: rows=%lld F=%d: %.3f ms, %.1f TF algorithmic, %.1f TF fp16 executed (of 2500), spectra %.2f TB/s   %s\n", rows, F, ms,
           fl / ms / 1e9, (double)((rows + 31) / 32) * F * 136 * 12 * 16384.0 / ms / 1e9, (double)rows * F * 2048 / ms * 1e-9,
           hipGetErrorString(hipGetLastError()));
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) k3<<<F, 512, spycsd::M3_LDS_BYTES>>>(b);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("csd3m: %.3f ms, %.1f TF algorithmic\n", ms, fl / ms / 1e9);
    return 0;
}
