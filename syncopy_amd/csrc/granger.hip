// Host side of K6: regularisation, Wilson iteration and Granger causality on the device
// (spyhip_granger of include/spyhip.h).  One scalar (the convergence error) is read back per
// iteration, as is one vector of F eigenvalue estimates per condition-number evaluation.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "spy_common.h"
#include "granger_kernels.h"
#include "wilson_plus_kernel.h"

using spywil::cd;

namespace {
const double PI = 3.14159265358979323846264338327950288;

struct Dev {
    std::vector<void*> ptrs;
    ~Dev() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T> T* alloc(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return reinterpret_cast<T*>(p);
    }
};

// matrix-core path only (n >= 48): Badd joins op(B); Ref / part: per-workgroup maxima of |Ref - C| / |Ref| instead of C
int gemm(spyhip_ctx* ctx, const cd* A, const cd* B, cd* C, int n, int batch, long long sA, long long sB, long long sC,
         int opB, int addI, const cd* Badd = nullptr, const cd* Ref = nullptr, double* part = nullptr) {
    if (n >= 48) {      // fp64 matrix cores, 64 x 64 tiles
        const int ntx = (n + spywil::MT - 1) / spywil::MT;
        dim3 grid((unsigned)(ntx * ntx * ((batch + 7) / 8) * 8));      // XCD-aware 1-D grid, see the kernel
        if (!part && !Badd && opB == 1 && A == B && sA == sB)               // X X^H: Hermitian product
            hipLaunchKernelGGL(spywil::zgemm_mfma_kernel<3>, grid, dim3(256), 0, ctx->stream, A, B, C, n, sA, sB, sC, opB, addI,
                               Badd, Ref, part, batch);
        else if (part)
            hipLaunchKernelGGL(spywil::zgemm_mfma_kernel<2>, grid, dim3(256), 0, ctx->stream, A, B, C, n, sA, sB, sC, opB, addI,
                               Badd, Ref, part, batch);
        else if (Badd)
            hipLaunchKernelGGL(spywil::zgemm_mfma_kernel<1>, grid, dim3(256), 0, ctx->stream, A, B, C, n, sA, sB, sC, opB, addI,
                               Badd, Ref, part, batch);
        else
            hipLaunchKernelGGL(spywil::zgemm_mfma_kernel<0>, grid, dim3(256), 0, ctx->stream, A, B, C, n, sA, sB, sC, opB, addI,
                               Badd, Ref, part, batch);
        SPY_HIP_CHECK(hipGetLastError());
        return 0;
    }
    dim3 grid((n + spywil::GT - 1) / spywil::GT, (n + spywil::GT - 1) / spywil::GT, batch);
    hipLaunchKernelGGL(spywil::zgemm_kernel, grid, dim3(256), 0, ctx->stream, A, B, C, n, sA, sB, sC, opB, addI);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

int check_info(spyhip_ctx* ctx, int* info_d, int batch, const char* what) {
    std::vector<int> h(batch);
    SPY_HIP_CHECK(hipMemcpyAsync(h.data(), info_d, batch * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < batch; ++b)
        if (h[b]) { spy::set_error("%s failed for matrix %d of %d", what, b, batch); return -6; }
    return 0;
}

// blocked = true: the block Gauss-Jordan kernel (pivots inside 16 x 16 diagonal blocks only; info = 2 where a
// tiny pivot showed up and the caller must repeat with blocked = false); false: partial pivoting, 16x the traffic
// `src`: invert src into M (out of place) instead of M in place
int invert(spyhip_ctx* ctx, cd* M, int n, int batch, int* info_d, bool blocked = false, const cd* src = nullptr) {
    static const bool old_inverse = std::getenv("SPYHIP_INVERSE_OLD") != nullptr;
    if (blocked && n >= 2 * spywil::ZM && !old_inverse) {      // matrix-core block Gauss-Jordan
        const int npad = ((n + spywil::ZM - 1) / spywil::ZM) * spywil::ZM;
        const size_t lds = ((size_t)spywil::ZM * (npad + 1) + spywil::ZM * (spywil::ZM + 1)) * sizeof(cd);
        if (lds <= ctx->lds_per_block) {
            SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spywil::zinv_mfma_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(spywil::zinv_mfma_kernel, dim3(batch), dim3(spywil::ZT), lds, ctx->stream, M, src, n, info_d);
            SPY_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    if (src) SPY_HIP_CHECK(hipMemcpyAsync(M, src, (size_t)batch * n * n * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
    if (blocked && n >= 2 * spywil::ZB) {
        const int npad = ((n + spywil::ZB - 1) / spywil::ZB) * spywil::ZB;
        const size_t lds = ((size_t)spywil::ZB * npad + spywil::ZB * spywil::ZB) * sizeof(cd);
        if (lds <= ctx->lds_per_block) {
            SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spywil::zinv_blocked_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(spywil::zinv_blocked_kernel, dim3(batch), dim3(256), lds, ctx->stream, M, n, info_d);
            SPY_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    const size_t lds = (size_t)n * (2 * sizeof(cd) + sizeof(int));
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spywil::zinv_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(spywil::zinv_kernel, dim3(batch), dim3(256), lds, ctx->stream, M, n, info_d);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

int cholesky(spyhip_ctx* ctx, cd* M, int n, int batch, int* info_d) {
    hipLaunchKernelGGL(spywil::zchol_kernel, dim3(batch), dim3(256), (size_t)n * sizeof(cd), ctx->stream, M, n, info_d);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

// max_f cond_2(A_f) for Hermitian A_f: |lambda|_max(A) * |lambda|_max(A^-1) by power iteration - on the EIGHTH powers:
// three squarings on the matrix cores (13 ms at 2049 x 256 x 256) make the iteration converge eight times faster
// (the ratio of the two largest eigenvalues is raised to the 8th power; ~100 ms per call before), and the 8th root
// divides the estimate's relative error by 8.  `w1`, `w2`: two more work arrays of the size of A.
int max_cond(spyhip_ctx* ctx, const cd* A, cd* work, cd* w1, cd* w2, int n, int F, double* lam_d, int* info_d, double* out) {
    const size_t lds = (size_t)2 * n * sizeof(cd);
    const long long nn = (long long)n * n;
    const int iters = 400;
    std::vector<double> h(2 * (size_t)F);
    std::vector<int> hi(F);
    auto power8 = [&](const cd* X, double* lam) -> int {
        if (gemm(ctx, X, X, w1, n, F, nn, nn, nn, 0, 0)) return -2;        // X^2
        if (gemm(ctx, w1, w1, w2, n, F, nn, nn, nn, 0, 0)) return -2;      // X^4
        if (gemm(ctx, w2, w2, w1, n, F, nn, nn, nn, 0, 0)) return -2;      // X^8
        hipLaunchKernelGGL(spywil::power_kernel, dim3(F), dim3(256), lds, ctx->stream, w1, n, iters, lam);
        SPY_HIP_CHECK(hipGetLastError());
        return 0;
    };
    if (power8(A, lam_d)) return -2;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (invert(ctx, work, n, F, info_d, attempt == 0, A)) return -2;
        if (power8(work, lam_d + F)) return -2;
        SPY_HIP_CHECK(hipMemcpyAsync(h.data(), lam_d, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SPY_HIP_CHECK(hipMemcpyAsync(hi.data(), info_d, F * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        bool retry = false;
        for (int f = 0; f < F; ++f) retry = retry || hi[f] == 2;
        if (!retry) break;
    }
    double m = 0.0;
    for (int f = 0; f < F; ++f) {
        double c = hi[f] ? INFINITY : std::pow(h[f], 0.125) * std::pow(h[F + f], 0.125);
        if (!(c == c)) c = INFINITY;
        m = std::max(m, c);
    }
    *out = m;
    return 0;
}

// plus operator for power-of-two lag-domain lengths 256 .. 4096 (wilson_plus_kernel.h); false: no such kernel
template <int LOG2L>
int launch_plus4(spyhip_ctx* ctx, const cd* g, int F, int n, const cd* tw, cd* gp, cd* g0) {
    using C = spywil::PCfg<LOG2L>;
    auto kern = spywil::plus4_kernel<LOG2L>;
    static bool attr_set = false;
    if (!attr_set) {
        SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)C::LDS_BYTES));
        attr_set = true;
    }
    const size_t nn = (size_t)n * n;
    hipLaunchKernelGGL(kern, dim3((unsigned)((nn + 3) / 4)), dim3(C::T), C::LDS_BYTES, ctx->stream, g, F, n, tw, gp, g0);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}
int plus4(spyhip_ctx* ctx, int L, const cd* g, int F, int n, const cd* tw, cd* gp, cd* g0) {
    switch (L) {
        case 256: return launch_plus4<8>(ctx, g, F, n, tw, gp, g0);
        case 512: return launch_plus4<9>(ctx, g, F, n, tw, gp, g0);
        case 1024: return launch_plus4<10>(ctx, g, F, n, tw, gp, g0);
        case 2048: return launch_plus4<11>(ctx, g, F, n, tw, gp, g0);
        case 4096: return launch_plus4<12>(ctx, g, F, n, tw, gp, g0);
        default: return 1;
    }
}

bool plus_plan(int L, spywil::PlusPlan* pl) {
    pl->L = L;
    int k = 0, n = L;
    static const int cand[] = {4, 2, 3, 5, 7, 11, 13};
    for (int c : cand)
        while (n % c == 0 && n > 1) { if (k >= spywil::PO_MAXFAC) return false; pl->radix[k++] = c; n /= c; }
    for (int p = 17; n > 1; p += 2)
        while (n % p == 0) { if (k >= spywil::PO_MAXFAC) return false; pl->radix[k++] = p; n /= p; }
    pl->nfac = k;
    return true;
}

}  // namespace

extern "C" int spyhip_granger(spyhip_ctx* ctx, const void* csd_d, int nfreq, int nchan, double rtol, int niter,
                              double cond_max, double eps_max, void* granger_d, void* H_d, void* Sigma_d,
                              double* info) {
    if (!ctx || !csd_d || !granger_d || !info) { spy::set_error("granger: null argument"); return -1; }
    if (nfreq < 3 || nchan < 1) { spy::set_error("granger: need nfreq >= 3 and nchan >= 1"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const int F = nfreq, n = nchan, L = 2 * (F - 1);
    const size_t nn = (size_t)n * n, tot = (size_t)F * nn;
    spywil::PlusPlan pl;
    if (!plus_plan(L, &pl) || (size_t)2 * L * sizeof(cd) > ctx->lds_per_block) {
        spy::set_error("granger: %d frequencies (lag-domain length %d) exceed the LDS FFT of the plus operator "
                       "(the reference documents its defaults for up to 5000 samples)", F, L);
        return -3;
    }
    Dev dev;
    cd* A = dev.alloc<cd>(tot);
    cd* U = dev.alloc<cd>(tot);
    cd* psi = dev.alloc<cd>(tot);
    cd* T1 = dev.alloc<cd>(tot);
    cd* T2 = dev.alloc<cd>(tot);
    cd* small = dev.alloc<cd>(7 * nn);        // g0, psi0, psi0 next, g0+S, Sigma, scratch, psi0 of iteration 0
    cd* tw = dev.alloc<cd>(L);
    double* lam = dev.alloc<double>(2 * (size_t)F);
    int* inf = dev.alloc<int>(F);
    const int nred = 1024;
    double* part = dev.alloc<double>(nred);
    const int mtiles = (n + spywil::MT - 1) / spywil::MT;
    double* bigpart = dev.alloc<double>((size_t)mtiles * mtiles * F);
    if (!A || !U || !psi || !T1 || !T2 || !small || !tw || !lam || !inf || !part || !bigpart) {
        spy::set_error("granger: out of device memory (%zu bytes per work array)", tot * sizeof(cd));
        return -2;
    }
    cd *g0 = small, *psi0 = small + nn, *psi0n = small + 2 * nn, *g0S = small + 3 * nn, *Sig = small + 4 * nn,
       *scr = small + 5 * nn, *scr2 = small + 6 * nn;
    {
        std::vector<cd> h(L);
        for (int m = 0; m < L; ++m) { const double a = -2.0 * PI * m / L; h[m] = make_double2(std::cos(a), std::sin(a)); }
        SPY_HIP_CHECK(hipMemcpyAsync(tw, h.data(), L * sizeof(cd), hipMemcpyHostToDevice, ctx->stream));
        SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    const unsigned eb = (unsigned)std::min<size_t>((tot + 255) / 256, 8192);
    const float2* csd = reinterpret_cast<const float2*>(csd_d);

    // ---- regularize_csd (wilson_sf.py:197-254)
    double cond0 = 0.0, factor = 0.0;
    hipLaunchKernelGGL(spywil::widen_kernel, dim3(eb), dim3(256), 0, ctx->stream, csd, A, n, (long long)tot, 0.0);
    if (max_cond(ctx, A, T1, T2, psi, n, F, lam, inf, &cond0)) return -2;
    if (!(cond0 < cond_max)) {
        factor = -1.0;
        const int nsteps = 15;
        for (int s = 0; s < nsteps; ++s) {
            const double e10 = -10.0 + (std::log10(eps_max) + 10.0) * s / (nsteps - 1);
            const double eps = std::pow(10.0, e10);
            hipLaunchKernelGGL(spywil::widen_kernel, dim3(eb), dim3(256), 0, ctx->stream, csd, A, n, (long long)tot, eps);
            double c = 0.0;
            if (max_cond(ctx, A, T1, T2, psi, n, F, lam, inf, &c)) return -2;
            if (c < cond_max) { factor = eps; break; }
        }
    }

    // ---- Wilson factorisation (wilson_sf.py:16-120)
    SPY_HIP_CHECK(hipMemcpyAsync(U, A, tot * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
    if (cholesky(ctx, U, n, F, inf)) return -2;
    if (int rc = check_info(ctx, inf, F, "Cholesky factorisation of the CSD (not positive definite)")) return rc;
    hipLaunchKernelGGL(spywil::gamma0_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream, A, F, n, scr);
    if (cholesky(ctx, scr, n, 1, inf)) return -2;
    if (int rc = check_info(ctx, inf, 1, "Cholesky factorisation of gamma_0 (not positive definite)")) return rc;
    hipLaunchKernelGGL(spywil::transpose_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream, scr, psi0, n);
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spywil::plus_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * L * sizeof(cd))));
    SPY_HIP_CHECK(hipMemcpyAsync(scr2, psi0, nn * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));   // keep psi0 of iteration 0
    bool converged = false;
    double err = INFINITY;
    std::vector<int> hinf(F);
    static const bool use_plus4 = std::getenv("SPYHIP_PLUS_OLD") == nullptr;
  for (int attempt = 0; attempt < 2 && !converged; ++attempt) {
    // attempt 0 inverts psi with the block Gauss-Jordan kernel; if one of its diagonal blocks was (nearly)
    // singular anywhere, the whole iteration restarts with the partially pivoted kernel
    const bool blocked = attempt == 0;
    bool tiny_pivot = false;
    SPY_HIP_CHECK(hipMemcpyAsync(psi0, scr2, nn * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(spywil::tile_kernel, dim3(eb), dim3(256), 0, ctx->stream, psi0, psi, F, n);
    SPY_HIP_CHECK(hipGetLastError());
    err = INFINITY;
    ctx->granger_iters = 0;
    for (int it = 0; it < niter; ++it) {
        ctx->granger_iters = it + 1;
        if (invert(ctx, T1, n, F, inf, blocked, psi)) return -2;                          // T1 = psi^-1
        SPY_HIP_CHECK(hipMemcpyAsync(hinf.data(), inf, F * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        if (gemm(ctx, T1, U, T2, n, F, nn, nn, nn, 0, 0)) return -2;                       // psi^-1 U
        if (gemm(ctx, T2, T2, T1, n, F, nn, nn, nn, 1, 1)) return -2;                      // g + I
        {                                                                                   // T2 = [g+I]^+
            const int prc = use_plus4 ? plus4(ctx, L, T1, F, n, tw, T2, g0) : 1;
            if (prc < 0) return prc;
            if (prc > 0)
                hipLaunchKernelGGL(spywil::plus_kernel, dim3((unsigned)nn), dim3(256), 2 * L * sizeof(cd), ctx->stream,
                                   T1, F, n, pl, tw, T2, g0);
        }
        const bool fused = n >= 48;               // the matrix-core gemm takes S and the error check along
        if (fused) {
            hipLaunchKernelGGL(spywil::skew_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream, g0, scr, g0S, n);
            SPY_HIP_CHECK(hipGetLastError());
            if (gemm(ctx, psi, T2, T1, n, F, nn, nn, nn, 0, 0, scr)) return -2;            // psi (g+ + S)
        } else {
            hipLaunchKernelGGL(spywil::add_S_kernel, dim3(eb), dim3(256), 0, ctx->stream, T2, g0, g0S, F, n);
            SPY_HIP_CHECK(hipGetLastError());
            if (gemm(ctx, psi, T2, T1, n, F, nn, nn, nn, 0, 0)) return -2;                 // psi (g+ + S)
        }
        std::swap(psi, T1);
        if (gemm(ctx, psi0, g0S, psi0n, n, 1, nn, nn, nn, 0, 0)) return -2;                // psi0 (g+_0 + S)
        std::swap(psi0, psi0n);
        std::vector<double> hp(fused ? 1 : nred);
        if (fused) {
            const int tiles = (n + spywil::MT - 1) / spywil::MT, nwg = tiles * tiles * F;
            if (gemm(ctx, psi, psi, nullptr, n, F, nn, nn, nn, 1, 0, nullptr, A, bigpart)) return -2;   // |A - psi psi^H| / |A|
            hipLaunchKernelGGL(spywil::maxred_kernel, dim3(1), dim3(256), 0, ctx->stream, bigpart, nwg, part);
        } else {
            if (gemm(ctx, psi, psi, T1, n, F, nn, nn, nn, 1, 0)) return -2;                // psi psi^H
            hipLaunchKernelGGL(spywil::relerr_kernel, dim3(nred), dim3(256), 0, ctx->stream, A, T1, (long long)tot, part);
        }
        SPY_HIP_CHECK(hipGetLastError());
        SPY_HIP_CHECK(hipMemcpyAsync(hp.data(), part, hp.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        err = 0.0;
        for (double v : hp) if (v > err || v != v) err = v;
        for (int f = 0; f < F; ++f) tiny_pivot = tiny_pivot || hinf[f] == 2;
        if (tiny_pivot) break;
        if (err < rtol) { converged = true; break; }
    }
    if (!tiny_pivot) break;
  }
    // ---- noise covariance, transfer function, Granger causality (wilson_sf.py:113-120, granger.py:53-77)
    if (gemm(ctx, psi0, psi0, Sig, n, 1, nn, nn, nn, 1, 0)) return -2;                     // psi0 psi0^T (psi0 is real)
    SPY_HIP_CHECK(hipMemcpyAsync(scr, psi0, nn * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
    if (invert(ctx, scr, n, 1, inf)) return -2;
    if (gemm(ctx, psi, scr, T1, n, F, nn, 0, nn, 0, 0)) return -2;                         // H = psi psi0^-1
    hipLaunchKernelGGL(spywil::granger_kernel, dim3(eb), dim3(256), 0, ctx->stream, A, T1, Sig, F, n,
                       reinterpret_cast<float*>(granger_d));
    SPY_HIP_CHECK(hipGetLastError());
    if (H_d) SPY_HIP_CHECK(hipMemcpyAsync(H_d, T1, tot * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
    if (Sigma_d) SPY_HIP_CHECK(hipMemcpyAsync(Sigma_d, Sig, nn * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    info[0] = converged ? 1.0 : 0.0;
    info[1] = err;
    info[2] = factor;
    info[3] = cond0;
    return 0;
}

extern "C" int spyhip_granger_last_iterations(const spyhip_ctx* ctx) { return ctx ? ctx->granger_iters : -1; }
