// Torch-free driver of the SECONDARY kernels of bench.py for rocprofv3 counter / kernel-trace passes (PMC collection
// crashes inside torch's own kernels on this image): the same plans, shapes and launch arguments as bench.py's
// `secondary` entries, on synthetic trials.
//   pmc_harness2 <mode> [trials] [reps]
//   c2      256 ch x 4096, 7 tapers, power + taper mean                (BASELINE configs[1])
//   c2f64   the same through spyhip_fft_plan_set_precision(plan, 1)
//   n2000 / n5000 / n3000 ...   the c2 shape at another trial length
//   conv    128 ch x 16384, 512-sample Hann windows, 50 % overlap, pow  (configs[3] i)
//   wav     128 ch x 16384, Morlet w0 = 6, 25 scales 4 .. 100 Hz, trial average (configs[3] ii)
// build: hipcc -O2 tools/pmc_harness2.cpp -Iinclude -Lsyncopy_amd -lspyhip -Wl,-rpath,$PWD/syncopy_amd -o gpurun_out/pmc_harness2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "spyhip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define SK(x) do { int r = (x); if (r) { fprintf(stderr, "%s -> %d: %s\n", #x, r, spyhip_last_error()); return 1; } } while (0)

static float* synth(int T, int N, int C) {
    // one random trial, replicated on the device (counters and durations do not depend on the values)
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> h((size_t)N * C);
    for (auto& v : h) v = nd(rng);
    float* d = nullptr;
    if (hipMalloc(&d, (size_t)T * N * C * 4) != hipSuccess) return nullptr;
    for (int t = 0; t < T; ++t)
        if (hipMemcpy(d + (size_t)t * N * C, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "c2";
    spyhip_ctx* ctx;
    SK(spyhip_ctx_create(0, &ctx));
    if (mode == "conv" || mode == "wav") {
        const int C = 128, N = 16384, T = argc > 2 ? atoi(argv[2]) : 200, reps = argc > 3 ? atoi(argv[3]) : 2;
        float* data = synth(T, N, C);
        if (!data) { fprintf(stderr, "out of memory\n"); return 1; }
        if (mode == "conv") {
            const int nperseg = 512, step = 256, nT = (N + step - 1) / step;
            std::vector<double> w(nperseg);
            double sum = 0;
            for (int n = 0; n < nperseg; ++n) { w[n] = 0.5 - 0.5 * std::cos(2 * M_PI * n / (nperseg - 1)); sum += w[n]; }
            for (auto& v : w) v *= std::sqrt(4.0 / 3.0) * std::sqrt(nperseg / sum);
            spyhip_fft_plan* plan;
            SK(spyhip_fft_plan_create(ctx, nperseg, nperseg, C, 1, w.data(), std::sqrt(2.0) / nperseg, 0, 0, nullptr, 0, SPYHIP_OUT_POW, 0, &plan));
            std::vector<int64_t> st((size_t)T * nT), lo(st.size()), hi(st.size());
            for (int t = 0; t < T; ++t)
                for (int s = 0; s < nT; ++s) {
                    st[(size_t)t * nT + s] = (int64_t)t * N + (int64_t)s * step - nperseg / 2;
                    lo[(size_t)t * nT + s] = (int64_t)t * N;
                    hi[(size_t)t * nT + s] = (int64_t)(t + 1) * N;
                }
            int64_t *dst, *dlo, *dhi;
            CK(hipMalloc(&dst, st.size() * 8)); CK(hipMalloc(&dlo, st.size() * 8)); CK(hipMalloc(&dhi, st.size() * 8));
            CK(hipMemcpy(dst, st.data(), st.size() * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(dlo, lo.data(), st.size() * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(dhi, hi.data(), st.size() * 8, hipMemcpyHostToDevice));
            void* out; CK(hipMalloc(&out, st.size() * (nperseg / 2 + 1) * C * 4));
            for (int r = 0; r < reps; ++r) SK(spyhip_fft_exec(plan, data, C, nullptr, dst, dlo, dhi, (int)st.size(), out));
            SK(spyhip_ctx_synchronize(ctx));
            printf("kernel %s; conv T=%d reps=%d segments=%zu\n", spyhip_fft_plan_kernel_name(plan), T, reps, st.size());
        } else {
            std::vector<double> scales(25);
            for (int i = 0; i < 25; ++i) scales[i] = (1.0 / (4.0 * (i + 1))) * (6 + std::sqrt(38.0)) / (4 * M_PI);
            spyhip_cwt_plan* plan;
            SK(spyhip_cwt_plan_create(ctx, N, C, 25, scales.data(), 1e-3, 6.0, 0, SPYHIP_OUT_POW, nullptr, N, &plan));
            std::vector<int64_t> st(T), hi(T);
            for (int t = 0; t < T; ++t) { st[t] = (int64_t)t * N; hi[t] = st[t] + N; }
            int64_t *dst, *dhi;
            CK(hipMalloc(&dst, T * 8)); CK(hipMalloc(&dhi, T * 8));
            CK(hipMemcpy(dst, st.data(), T * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(dhi, hi.data(), T * 8, hipMemcpyHostToDevice));
            void* out; CK(hipMalloc(&out, (size_t)N * 25 * C * 4));
            CK(hipMemset(out, 0, (size_t)N * 25 * C * 4));
            for (int r = 0; r < reps; ++r) SK(spyhip_cwt_exec(plan, data, C, nullptr, dst, dst, dhi, T, out, 2));
            SK(spyhip_ctx_synchronize(ctx));
            printf("wav T=%d reps=%d\n", T, reps);
        }
        return 0;
    }
    // ---- c2 family: power spectra with taper mean, 256 channels, 7 tapers
    int N = 4096;
    bool f64 = false;
    if (mode == "c2f64") f64 = true;
    else if (mode[0] == 'n') { N = atoi(mode.c_str() + 1); const char* p = strstr(mode.c_str(), "f64"); f64 = p != nullptr; }
    const int C = 256, K = 7, T = argc > 2 ? atoi(argv[2]) : (N == 4096 ? 1000 : 200), reps = argc > 3 ? atoi(argv[3]) : 2;
    float* data = synth(T, N, C);
    if (!data) { fprintf(stderr, "out of memory\n"); return 1; }
    std::vector<double> tp((size_t)K * N);
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) tp[(size_t)k * N + n] = std::sin(M_PI * (k + 1) * (n + 0.5) / N) * std::sqrt(2.0);
    spyhip_fft_plan* plan;
    SK(spyhip_fft_plan_create(ctx, N, N, C, K, tp.data(), std::sqrt(2.0) / N, 0, 0, nullptr, 0, SPYHIP_OUT_POW, 0, &plan));
    SK(spyhip_fft_plan_set_reference_mean(plan, 1));
    if (f64) SK(spyhip_fft_plan_set_precision(plan, 1));
    std::vector<int64_t> st(T), hi(T);
    for (int t = 0; t < T; ++t) { st[t] = (int64_t)t * N; hi[t] = st[t] + N; }
    int64_t *dst, *dhi;
    CK(hipMalloc(&dst, T * 8)); CK(hipMalloc(&dhi, T * 8));
    CK(hipMemcpy(dst, st.data(), T * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dhi, hi.data(), T * 8, hipMemcpyHostToDevice));
    void* out; CK(hipMalloc(&out, (size_t)T * (N / 2 + 1) * C * 4));
    for (int r = 0; r < reps; ++r) SK(spyhip_fft_exec(plan, data, C, nullptr, dst, dst, dhi, T, out));
    SK(spyhip_ctx_synchronize(ctx));
    printf("kernel %s; %s N=%d T=%d reps=%d\n", spyhip_fft_plan_kernel_name(plan), mode.c_str(), N, T, reps);
    return 0;
}
