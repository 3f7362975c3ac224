// Torch-free driver of the C ABI for rocprofv3 counter passes (PMC collection crashes inside
// torch's own kernels on this image).  Runs the coherence front half on synthetic trials:
//   spyhip_fft_exec (fourier, all tapers; leaves the range of the spectra) -> spyhip_csd_accumulate_split (K4h on the
//   half-precision matrix cores; SPYHIP_CSD_F32=1: the float32 kernels) or spyhip_csd_accumulate_blocked, `reps` times;
//   which & 4: the trials are 64 distinct AR(2) realisations (alphas 0.55, -0.8 as synthdata.ar2_network), and the
//   averaged CSD goes through spyhip_granger afterwards (K6 at 256 channels x 2049 frequencies).
// build: hipcc -O2 tools/pmc_harness.cpp -Iinclude -Lsyncopy_amd -lspyhip -Wl,-rpath,$PWD/syncopy_amd -o gpurun_out/pmc_harness
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "spyhip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define SK(x) do { int r = (x); if (r) { fprintf(stderr, "%s -> %d: %s\n", #x, r, spyhip_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 500, reps = argc > 2 ? atoi(argv[2]) : 2;
    const int which = argc > 3 ? atoi(argv[3]) : 3;   // bit 0: fft, bit 1: csd, bit 2: granger on the result
    const int blocked = argc > 4 ? atoi(argv[4]) : 0; // 1: channel-blocked hand-over layout (bench.py --blocked)
    const int C = 256, N = 4096, K = 7, F = N / 2 + 1;
    spyhip_ctx* ctx;
    SK(spyhip_ctx_create(0, &ctx));
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    // one random trial, replicated on the device (counter collection does not care about the values)
    const int ndistinct = (which & 4) ? (B < 64 ? B : 64) : 1;
    std::vector<float> h((size_t)ndistinct * N * C);
    for (auto& v : h) v = nd(rng);
    if (which & 4)
        for (int t = 0; t < ndistinct; ++t)
            for (int n = 2; n < N; ++n)
                for (int c = 0; c < C; ++c) {
                    float* x = h.data() + (size_t)t * N * C;
                    x[(size_t)n * C + c] += 0.55f * x[(size_t)(n - 1) * C + c] - 0.8f * x[(size_t)(n - 2) * C + c];
                }
    const size_t tl = (size_t)N * C;
    float* data; CK(hipMalloc(&data, (size_t)B * tl * 4));
    for (int b = 0; b < B; ++b) CK(hipMemcpy(data + (size_t)b * tl, h.data() + (size_t)(b % ndistinct) * tl, tl * 4, hipMemcpyHostToDevice));
    std::vector<int64_t> st(B), hi(B);
    for (int b = 0; b < B; ++b) { st[b] = (int64_t)b * N; hi[b] = st[b] + N; }
    int64_t *dst, *dhi; CK(hipMalloc(&dst, B * 8)); CK(hipMalloc(&dhi, B * 8));
    CK(hipMemcpy(dst, st.data(), B * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dhi, hi.data(), B * 8, hipMemcpyHostToDevice));
    // any smooth windows do for counter collection (the DPSS tables come from SciPy on the Python side)
    std::vector<double> tp((size_t)K * N);
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) tp[(size_t)k * N + n] = std::sin(M_PI * (k + 1) * (n + 0.5) / N) * std::sqrt(2.0);
    spyhip_fft_plan* plan;
    SK(spyhip_fft_plan_create(ctx, N, N, C, K, tp.data(), std::sqrt(2.0) / N, 0, 0, nullptr, 0, SPYHIP_OUT_FOURIER, 1, &plan));
    SK(spyhip_fft_plan_set_reference_mean(plan, 1));     // as the front ends do for whole trials
    if (blocked) SK(spyhip_fft_plan_set_blocked(plan, 1));
    void *spec, *acc;
    float* absmax;
    CK(hipMalloc(&absmax, C * sizeof(float)));
    CK(hipMemset(absmax, 0, C * sizeof(float)));
    const bool ranged = !blocked && spyhip_fft_plan_set_absmax(plan, absmax) == 0;
    CK(hipMalloc(&spec, (size_t)B * K * F * C * 8));
    CK(hipMalloc(&acc, (size_t)F * C * C * 8));
    CK(hipMemset(acc, 0, (size_t)F * C * C * 8));
    CK(hipMemset(spec, 0, (size_t)B * K * F * C * 8));
    for (int r = 0; r < reps; ++r) {
        if (which & 1) SK(spyhip_fft_exec(plan, data, C, nullptr, dst, dst, dhi, B, spec));
        if (which & 2) SK(blocked ? spyhip_csd_accumulate_blocked(ctx, spec, (int64_t)B * K, F, C, acc)
                               : spyhip_csd_accumulate_split(ctx, spec, (int64_t)B * K, F, C, acc, ranged ? absmax : nullptr));
    }
    if ((which & 2) && !blocked) {
        int nfb = 0;
        SK(spyhip_csd_split_fallbacks(ctx, &nfb));
        printf("frequencies left to the float32 kernels: %d\n", nfb);
    }
    if (which & 4) {
        SK(spyhip_csd_finalize(ctx, acc, F, C, 1.0 / ((double)B * K * reps)));
        void* gr; CK(hipMalloc(&gr, (size_t)F * C * C * 4));
        double info[4];
        SK(spyhip_granger(ctx, acc, F, C, 5e-6, 100, 1e4, 1e-1, gr, nullptr, nullptr, info));
        printf("granger: converged %g, max rel. err %g, reg. factor %g, initial cond. num %g, %d iterations\n", info[0], info[1],
               info[2], info[3], spyhip_granger_last_iterations(ctx));
    }
    SK(spyhip_ctx_synchronize(ctx));
    printf("kernel %s; done B=%d reps=%d\n", spyhip_fft_plan_kernel_name(plan), B, reps);
    return 0;
}
