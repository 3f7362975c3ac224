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

// Work arrays of one spyhip_granger call, carved out of the context's arena (grown on demand, kept between calls:
// allocating and freeing 11 GB per call cost between 0.05 and 1 s at 256 channels x 2049 frequencies)
struct Dev {
    spyhip_ctx* ctx;
    size_t off = 0;
    bool ok = true;
    Dev(spyhip_ctx* c, size_t need) : ctx(c) {
        if (need > ctx->arena_bytes) {
            if (ctx->arena) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->arena); ctx->arena = nullptr; ctx->arena_bytes = 0; }
            if (hipMalloc(&ctx->arena, need) != hipSuccess) { ok = false; return; }
            ctx->arena_bytes = need;
        }
    }
    template <typename T> T* alloc(size_t n) {
        const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        if (!ok || off + bytes > ctx->arena_bytes) { ok = false; return nullptr; }
        T* p = reinterpret_cast<T*>(static_cast<char*>(ctx->arena) + off);
        off += bytes;
        return p;
    }
};

// matrix-core path only (n >= 48): Badd joins op(B); Ref / part: per-workgroup maxima of |Ref - C| / |Ref| instead of C
int gemm(spyhip_ctx* ctx, const cd* A, const cd* B, cd* C, int n, int batch, long long sA, long long sB, long long sC,
         int opB, int addI, const cd* Badd = nullptr, const cd* Ref = nullptr, double* part = nullptr) {
    if (n >= 48) {      // fp64 matrix cores, 64 x 64 tiles
        const int mode = part ? 2 : (Badd ? 1 : ((opB == 1 && A == B && sA == sB) ? 3 : 0));     // 3: X X^H, Hermitian product
        dim3 grid((unsigned)(spywil::zgemm_groups(n, mode) * ((batch + 7) / 8) * 8));            // XCD-aware 1-D grid, see the kernel
        if (mode == 3)
            hipLaunchKernelGGL(spywil::zgemm_mfma_kernel<3>, grid, dim3(256), 0, ctx->stream, A, B, C, n, sA, sB, sC, opB, addI,
                               Badd, Ref, part, batch);
        else if (mode == 2)
            hipLaunchKernelGGL(spywil::zgemm_mfma_kernel<2>, grid, dim3(256), 0, ctx->stream, A, B, C, n, sA, sB, sC, opB, addI,
                               Badd, Ref, part, batch);
        else if (mode == 1)
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
    // 64-row blocks (half the sweeps over the matrices) where they pad no more than the 32-row blocks would
    if (blocked && n >= 2 * spywil::ZW && (n + 63) / 64 * 64 == (n + 31) / 32 * 32) {
        const size_t lds = (size_t)2 * spywil::ZW * (spywil::ZW + 1) * sizeof(cd);
        SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spywil::zinv64_mfma_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(spywil::zinv64_mfma_kernel, dim3(batch), dim3(spywil::ZT), lds, ctx->stream, M, src, n, info_d);
        SPY_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (blocked && n >= 2 * spywil::ZM) {      // matrix-core block Gauss-Jordan
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

// inverse of ONE matrix (psi0): block Gauss-Jordan first (the pivoted kernel is a 256-step serial chain, 17.6 ms at
// n = 256 against < 1 ms), the pivoted kernel only if a diagonal block met a tiny pivot
int invert_one(spyhip_ctx* ctx, cd* dst, const cd* src, int n, int* inf) {
    if (invert(ctx, dst, n, 1, inf, true, src)) return -2;
    int h = 0;
    SPY_HIP_CHECK(hipMemcpyAsync(&h, inf, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (h == 0) return 0;
    return invert(ctx, dst, n, 1, inf, false, src);
}

int cholesky(spyhip_ctx* ctx, cd* M, int n, int batch, int* info_d) {
    const size_t plds = ((size_t)n * (spywil::CHP + 1) + spywil::CHP * (spywil::CHP + 1)) * sizeof(cd);
    if (n <= 256 && n >= 2 * spywil::CHP && plds <= ctx->lds_per_block) {      // panels of 32 columns
        SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spywil::zchol_panel_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
        hipLaunchKernelGGL(spywil::zchol_panel_kernel, dim3(batch), dim3(256), plds, ctx->stream, M, n, info_d);
        SPY_HIP_CHECK(hipGetLastError());
        return 0;
    }
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
        // (X is Hermitian: X^2 = X X^H, the product that computes the lower-triangle tiles only)
        if (gemm(ctx, X, X, w1, n, F, nn, nn, nn, 1, 0)) return -2;        // X^2
        if (gemm(ctx, w1, w1, w2, n, F, nn, nn, nn, 1, 0)) return -2;      // X^4
        if (gemm(ctx, w2, w2, w1, n, F, nn, nn, nn, 1, 0)) return -2;      // X^8
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
int launch_plus4(spyhip_ctx* ctx, const cd* g, int F, long long nent, const cd* tw, cd* gp, cd* g0) {
    using C = spywil::PCfg<LOG2L>;
    auto kern = spywil::plus4_kernel<LOG2L>;
    // (per device, cheap: set at every launch)
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3((unsigned)spywil::plus4_grid(nent)), dim3(C::T), C::LDS_BYTES, ctx->stream, g, F, nent, tw, gp, g0);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}
int plus4(spyhip_ctx* ctx, int L, const cd* g, int F, long long nent, const cd* tw, cd* gp, cd* g0) {
    switch (L) {
        case 256: return launch_plus4<8>(ctx, g, F, nent, tw, gp, g0);
        case 512: return launch_plus4<9>(ctx, g, F, nent, tw, gp, g0);
        case 1024: return launch_plus4<10>(ctx, g, F, nent, tw, gp, g0);
        case 2048: return launch_plus4<11>(ctx, g, F, nent, tw, gp, g0);
        case 4096: return launch_plus4<12>(ctx, g, F, nent, tw, gp, g0);
        default: return 1;
    }
}

using spywil::plus_plan;          // f64_stockham.h

// [g]^+ for nent entries over F rfft bins: the radix-16 LDS kernel for power-of-two lag-domain lengths 256 ... 4096,
// the generic LDS kernel while two length-L arrays fit LDS, global scratch beyond (any length)
int plus_any(spyhip_ctx* ctx, int L, const spywil::PlusPlan& pl, const cd* g, int F, long long nent, const cd* tw, cd* gp, cd* g0,
             bool use_plus4 = true) {
    const int prc = use_plus4 ? plus4(ctx, L, g, F, nent, tw, gp, g0) : 1;
    if (prc <= 0) return prc;
    const size_t lds = (size_t)2 * L * sizeof(cd);
    if (lds <= ctx->lds_per_block) {
        SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spywil::plus_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(spywil::plus_kernel, dim3((unsigned)nent), dim3(256), lds, ctx->stream, g, F, nent, pl, tw, gp, g0);
        SPY_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // entries per launch: scratch of at most 1 GiB (at least one workgroup per CU if that is more)
    long long chunk = std::max<long long>(ctx->num_cu, ((size_t)1 << 30) / lds);
    if (chunk > nent) chunk = nent;
    const size_t need = (size_t)chunk * lds;
    if (need > ctx->scratch_bytes) {
        if (ctx->scratch) { SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->scratch); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
        SPY_HIP_CHECK(hipMalloc(&ctx->scratch, need));
        ctx->scratch_bytes = need;
    }
    for (long long e0 = 0; e0 < nent; e0 += chunk) {
        const long long ne = std::min(chunk, nent - e0);
        hipLaunchKernelGGL(spywil::plus_long_kernel, dim3((unsigned)ne), dim3(256), 0, ctx->stream, g, F, nent, pl, tw, gp, g0,
                           reinterpret_cast<cd*>(ctx->scratch), e0);
    }
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
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
    if (!plus_plan(L, &pl)) {
        spy::set_error("granger: no radix schedule for the lag-domain length %d (%d frequencies)", L, F);
        return -3;
    }
    const int mtiles_ = (n + spywil::MT - 1) / spywil::MT;
    Dev dev(ctx, (5 * tot + 7 * nn + (size_t)L) * sizeof(cd) + (2 * (size_t)F + 1024 + (size_t)mtiles_ * mtiles_ * F) * sizeof(double) +
                     (size_t)F * sizeof(int) + 16 * 256);
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
    hipLaunchKernelGGL(spywil::gamma0_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream, A, F, n, scr, 0, F);
    if (cholesky(ctx, scr, n, 1, inf)) return -2;
    if (int rc = check_info(ctx, inf, 1, "Cholesky factorisation of gamma_0 (not positive definite)")) return rc;
    hipLaunchKernelGGL(spywil::transpose_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream, scr, psi0, n);
    SPY_HIP_CHECK(hipMemcpyAsync(scr2, psi0, nn * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));   // keep psi0 of iteration 0
    bool converged = false;
    double err = INFINITY;
    bool subset_only = false;            // the last error came from the frequency subset only (a lower bound)
    std::vector<int> hinf(F);
    constexpr bool use_plus4 = true;
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
        if (gemm(ctx, T1, U, T2, n, F, nn, nn, nn, 0, n >= 48 ? 2 : 0)) return -2;          // psi^-1 U (U lower triangular)
        if (gemm(ctx, T2, T2, T1, n, F, nn, nn, nn, 1, 1)) return -2;                      // g + I
        if (int prc = plus_any(ctx, L, pl, T1, F, (long long)nn, tw, T2, g0, use_plus4)) return prc;      // T2 = [g+I]^+
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
            // max_rel_err(CSD, psi psi^H) (wilson_sf.py:99-103,190-194) decides whether the loop stops.  The maximum over a
            // SUBSET of the frequencies is a lower bound of it: while every 8th bin alone is still above rtol the iteration
            // cannot have converged and the other 7/8 of the product need not be formed; the full check runs as soon as
            // the subset passes (and once more if the loop ends unconverged, for the reported error) - same decisions,
            // same reported value, ~1/8 of the 3.1 ms per iteration at 256 channels x 2049 frequencies.
            const int Fs = (F + 7) / 8;
            subset_only = false;
            if (F >= 64 && !std::getenv("SPYHIP_WILSON_FULL_CHECK")) {
                if (gemm(ctx, psi, psi, nullptr, n, Fs, 8 * (long long)nn, 8 * (long long)nn, 8 * (long long)nn, 1, 0, nullptr, A, bigpart)) return -2;
                hipLaunchKernelGGL(spywil::maxred_kernel, dim3(1), dim3(256), 0, ctx->stream, bigpart, spywil::zgemm_tiles(n, true) * Fs, part);
                SPY_HIP_CHECK(hipMemcpyAsync(hp.data(), part, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                subset_only = hp[0] >= rtol || hp[0] != hp[0];
            }
            if (!subset_only) {
                const int nwg = spywil::zgemm_tiles(n, true) * F;
                if (gemm(ctx, psi, psi, nullptr, n, F, nn, nn, nn, 1, 0, nullptr, A, bigpart)) return -2;   // |A - psi psi^H| / |A|
                hipLaunchKernelGGL(spywil::maxred_kernel, dim3(1), dim3(256), 0, ctx->stream, bigpart, nwg, part);
            }
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
    if (!converged && subset_only && n >= 48) {      // the loop ran out of iterations: report the error over ALL frequencies
        double full = 0.0;
        if (gemm(ctx, psi, psi, nullptr, n, F, nn, nn, nn, 1, 0, nullptr, A, bigpart)) return -2;
        hipLaunchKernelGGL(spywil::maxred_kernel, dim3(1), dim3(256), 0, ctx->stream, bigpart, spywil::zgemm_tiles(n, true) * F, part);
        SPY_HIP_CHECK(hipMemcpyAsync(&full, part, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        err = full;
    }
    // ---- noise covariance, transfer function, Granger causality (wilson_sf.py:113-120, granger.py:53-77)
    if (gemm(ctx, psi0, psi0, Sig, n, 1, nn, nn, nn, 1, 0)) return -2;                     // psi0 psi0^T (psi0 is real)
    if (invert_one(ctx, scr, psi0, n, inf)) return -2;
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

// =====================================================================================================================
// The Wilson factorisation in steps, for frequency shards: every rank holds the bins [f_lo, f_lo + nf) of
// the nftot rfft bins of the CSD.  Everything per frequency (regularisation, Cholesky factor, inverse, products, error)
// is local; the plus operator works along the frequency axis, so the host transposes g = psi^-1 S psi^-H between
// "frequency shards x all entries" and "all frequencies x entry shards" around spyhip_wilson_plus (an all-to-all),
// and sums / maximises three small quantities over ranks (gamma_0, the condition number, the error).  The host side
// is syncopy_amd/connectivity/wilson_sharded.py; with one rank the sequence equals spyhip_granger.
// All arrays are complex128 on the device unless stated; work_d: 3 x nf x n x n complex128.
// =====================================================================================================================
namespace {
struct Tmp {        // small per-call device scratch
    void* p = nullptr;
    ~Tmp() { if (p) (void)hipFree(p); }
    int get(size_t bytes) { return hipMalloc(&p, bytes) == hipSuccess ? 0 : -2; }
};
}  // namespace

// A = widen(csd) + eps I on the local bins (regularize_csd's CSD + eps*eye, wilson_sf.py:244); cond_out (host): the
// largest 2-norm condition number of the local bins.  work_d as above.
extern "C" int spyhip_wilson_cond(spyhip_ctx* ctx, const void* csd_c64_d, int nf, int n, double eps, void* A_d, void* work_d,
                                  double* cond_out) {
    if (!ctx || !csd_c64_d || !A_d || !work_d || !cond_out) { spy::set_error("wilson_cond: null argument"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t tot = (size_t)nf * n * n;
    cd* A = reinterpret_cast<cd*>(A_d);
    cd* W = reinterpret_cast<cd*>(work_d);
    Tmp t;
    if (t.get(2 * (size_t)nf * sizeof(double) + nf * sizeof(int))) { spy::set_error("wilson_cond: out of device memory"); return -2; }
    double* lam = reinterpret_cast<double*>(t.p);
    int* inf = reinterpret_cast<int*>(lam + 2 * (size_t)nf);
    const unsigned eb = (unsigned)std::min<size_t>((tot + 255) / 256, 8192);
    hipLaunchKernelGGL(spywil::widen_kernel, dim3(eb), dim3(256), 0, ctx->stream, reinterpret_cast<const float2*>(csd_c64_d), A, n,
                       (long long)tot, eps);
    SPY_HIP_CHECK(hipGetLastError());
    return max_cond(ctx, A, W, W + tot, W + 2 * tot, n, nf, lam, inf, cond_out);
}

// U = Cholesky factor of A per local bin (wilson_sf.py:76) and this shard's part of gamma_0 = fft(CSD_full)[0]
// (wilson_sf.py:135-140, symmetrised real part): gamma_part_d (n x n) is to be summed over ranks.
extern "C" int spyhip_wilson_init(spyhip_ctx* ctx, const void* A_d, int nf, int n, int f_lo, int nftot, void* U_d,
                                  void* gamma_part_d) {
    if (!ctx || !A_d || !U_d || !gamma_part_d) { spy::set_error("wilson_init: null argument"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n, tot = (size_t)nf * nn;
    Tmp t;
    if (t.get((size_t)nf * sizeof(int))) return -2;
    int* inf = reinterpret_cast<int*>(t.p);
    SPY_HIP_CHECK(hipMemcpyAsync(U_d, A_d, tot * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
    if (cholesky(ctx, reinterpret_cast<cd*>(U_d), n, nf, inf)) return -2;
    if (int rc = check_info(ctx, inf, nf, "Cholesky factorisation of the CSD (not positive definite)")) return rc;
    hipLaunchKernelGGL(spywil::gamma0_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const cd*>(A_d), nf, n, reinterpret_cast<cd*>(gamma_part_d), f_lo, nftot);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

// psi0 = chol(gamma_0)^T (wilson_sf.py:144-151) from the summed gamma_0, tiled over the local bins into psi_d
extern "C" int spyhip_wilson_psi0(spyhip_ctx* ctx, void* gamma0_d, int n, int nf, void* psi0_d, void* psi_d) {
    if (!ctx || !gamma0_d || !psi0_d || !psi_d) { spy::set_error("wilson_psi0: null argument"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n, tot = (size_t)nf * nn;
    Tmp t;
    if (t.get(sizeof(int))) return -2;
    int* inf = reinterpret_cast<int*>(t.p);
    if (cholesky(ctx, reinterpret_cast<cd*>(gamma0_d), n, 1, inf)) return -2;
    if (int rc = check_info(ctx, inf, 1, "Cholesky factorisation of gamma_0 (not positive definite)")) return rc;
    hipLaunchKernelGGL(spywil::transpose_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const cd*>(gamma0_d), reinterpret_cast<cd*>(psi0_d), n);
    const unsigned eb = (unsigned)std::min<size_t>((tot + 255) / 256, 8192);
    hipLaunchKernelGGL(spywil::tile_kernel, dim3(eb), dim3(256), 0, ctx->stream, reinterpret_cast<const cd*>(psi0_d),
                       reinterpret_cast<cd*>(psi_d), nf, n);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

// g = (psi^-1 U)(psi^-1 U)^H + I on the local bins (wilson_sf.py:80-92).  work_d: 2 x nf x n x n.  Returns 1 (not an
// error) if the block inverse met a tiny pivot: repeat the whole factorisation with pivoted = 1.
extern "C" int spyhip_wilson_g(spyhip_ctx* ctx, const void* psi_d, const void* U_d, int nf, int n, int pivoted, void* work_d,
                               void* g_d) {
    if (!ctx || !psi_d || !U_d || !work_d || !g_d) { spy::set_error("wilson_g: null argument"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const long long nn = (long long)n * n;
    const size_t tot = (size_t)nf * nn;
    cd* T1 = reinterpret_cast<cd*>(work_d);
    cd* T2 = T1 + tot;
    Tmp t;
    if (t.get((size_t)nf * sizeof(int))) return -2;
    int* inf = reinterpret_cast<int*>(t.p);
    if (invert(ctx, T1, n, nf, inf, !pivoted, reinterpret_cast<const cd*>(psi_d))) return -2;
    std::vector<int> h(nf);
    SPY_HIP_CHECK(hipMemcpyAsync(h.data(), inf, nf * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (gemm(ctx, T1, reinterpret_cast<const cd*>(U_d), T2, n, nf, nn, nn, nn, 0, n >= 48 ? 2 : 0)) return -2;
    if (gemm(ctx, T2, T2, reinterpret_cast<cd*>(g_d), n, nf, nn, nn, nn, 1, 1)) return -2;
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int f = 0; f < nf; ++f)
        if (h[f] == 2) return 1;
    return 0;
}

// plus operator (wilson_sf.py:154-184) for nent matrix entries over ALL nftot frequencies: g_d, gp_d (nftot, nent),
// g0_d (nent): the zero-lag coefficients (halved, real).
extern "C" int spyhip_wilson_plus(spyhip_ctx* ctx, const void* g_d, int nftot, int64_t nent, void* gp_d, void* g0_d) {
    if (!ctx || !g_d || !gp_d || !g0_d) { spy::set_error("wilson_plus: null argument"); return -1; }
    if (nent <= 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const int L = 2 * (nftot - 1);
    spywil::PlusPlan pl;
    if (!plus_plan(L, &pl)) {
        spy::set_error("wilson_plus: no radix schedule for the lag-domain length %d", L);
        return -3;
    }
    Tmp t;
    if (t.get((size_t)L * sizeof(cd))) return -2;
    cd* tw = reinterpret_cast<cd*>(t.p);
    std::vector<cd> h(L);
    for (int m = 0; m < L; ++m) { const double a = -2.0 * PI * m / L; h[m] = make_double2(std::cos(a), std::sin(a)); }
    SPY_HIP_CHECK(hipMemcpyAsync(tw, h.data(), L * sizeof(cd), hipMemcpyHostToDevice, ctx->stream));
    if (int prc = plus_any(ctx, L, pl, reinterpret_cast<const cd*>(g_d), nftot, (long long)nent, tw, reinterpret_cast<cd*>(gp_d),
                           reinterpret_cast<cd*>(g0_d)))
        return prc;
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));        // tw is freed on return
    return 0;
}

// psi <- psi (g+ + S), psi0 <- psi0 (g0 + S) with S = triu(g0) - triu(g0)^H (wilson_sf.py:97-101); err_out (host): this
// shard's max |A - psi psi^H| / |A| (:103, :190-194).  g0_d (n x n) holds ALL entries (gathered by the host).
// work_d: nf x n x n.
extern "C" int spyhip_wilson_update(spyhip_ctx* ctx, void* psi_d, const void* gp_d, const void* g0_d, void* psi0_d,
                                    const void* A_d, int nf, int n, void* work_d, double* err_out) {
    if (!ctx || !psi_d || !gp_d || !g0_d || !psi0_d || !A_d || !work_d || !err_out) { spy::set_error("wilson_update: null argument"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const long long nn = (long long)n * n;
    const size_t tot = (size_t)nf * nn;
    cd* psi = reinterpret_cast<cd*>(psi_d);
    cd* T1 = reinterpret_cast<cd*>(work_d);
    const int mt = (n + spywil::MT - 1) / spywil::MT;
    const size_t npart = std::max<size_t>((size_t)mt * mt * nf, 1024);
    Tmp t;
    if (t.get(3 * (size_t)nn * sizeof(cd) + (npart + 1) * sizeof(double))) return -2;
    cd* S = reinterpret_cast<cd*>(t.p);
    cd* g0S = S + nn;
    cd* p0n = g0S + nn;
    double* part = reinterpret_cast<double*>(p0n + nn);
    hipLaunchKernelGGL(spywil::skew_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const cd*>(g0_d), S, g0S, n);
    SPY_HIP_CHECK(hipGetLastError());
    const bool fused = n >= 48;
    if (fused) {
        if (gemm(ctx, psi, reinterpret_cast<const cd*>(gp_d), T1, n, nf, nn, nn, nn, 0, 0, S)) return -2;
    } else {
        // small matrices: g+ + S in a pass of its own, into the work array (gp_d is left alone)
        SPY_HIP_CHECK(hipMemcpyAsync(T1, gp_d, tot * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
        const unsigned eb = (unsigned)std::min<size_t>((tot + 255) / 256, 8192);
        hipLaunchKernelGGL(spywil::add_S_kernel, dim3(eb), dim3(256), 0, ctx->stream, T1, reinterpret_cast<const cd*>(g0_d), g0S, nf, n);
        Tmp t2;
        if (t2.get(tot * sizeof(cd))) return -2;
        if (gemm(ctx, psi, T1, reinterpret_cast<cd*>(t2.p), n, nf, nn, nn, nn, 0, 0)) return -2;
        SPY_HIP_CHECK(hipMemcpyAsync(T1, t2.p, tot * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
        SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    SPY_HIP_CHECK(hipMemcpyAsync(psi, T1, tot * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
    if (gemm(ctx, reinterpret_cast<cd*>(psi0_d), g0S, p0n, n, 1, nn, nn, nn, 0, 0)) return -2;
    SPY_HIP_CHECK(hipMemcpyAsync(psi0_d, p0n, nn * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
    double* outp = part + npart;
    if (fused) {
        if (gemm(ctx, psi, psi, nullptr, n, nf, nn, nn, nn, 1, 0, nullptr, reinterpret_cast<const cd*>(A_d), part)) return -2;
        hipLaunchKernelGGL(spywil::maxred_kernel, dim3(1), dim3(256), 0, ctx->stream, part, spywil::zgemm_tiles(n, true) * nf, outp);
    } else {
        if (gemm(ctx, psi, psi, T1, n, nf, nn, nn, nn, 1, 0)) return -2;
        hipLaunchKernelGGL(spywil::relerr_kernel, dim3(1024), dim3(256), 0, ctx->stream, reinterpret_cast<const cd*>(A_d), T1,
                           (long long)tot, part);
        hipLaunchKernelGGL(spywil::maxred_kernel, dim3(1), dim3(256), 0, ctx->stream, part, 1024, outp);
    }
    SPY_HIP_CHECK(hipGetLastError());
    SPY_HIP_CHECK(hipMemcpyAsync(err_out, outp, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// Sigma = psi0 psi0^T, H = psi psi0^-1, Granger-Geweke causality on the local bins (wilson_sf.py:113-120,
// granger.py:53-77).  granger_d float32 (nf, n, n); H_d (nf, n, n) / Sigma_d (n, n) complex128 may be NULL.
extern "C" int spyhip_wilson_finish(spyhip_ctx* ctx, const void* A_d, const void* psi_d, const void* psi0_d, int nf, int n,
                                    void* work_d, void* granger_d, void* H_d, void* Sigma_d) {
    if (!ctx || !A_d || !psi_d || !psi0_d || !work_d || !granger_d) { spy::set_error("wilson_finish: null argument"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const long long nn = (long long)n * n;
    const size_t tot = (size_t)nf * nn;
    cd* T1 = reinterpret_cast<cd*>(work_d);
    Tmp t;
    if (t.get(2 * (size_t)nn * sizeof(cd) + sizeof(int))) return -2;
    cd* Sig = reinterpret_cast<cd*>(t.p);
    cd* inv0 = Sig + nn;
    int* inf = reinterpret_cast<int*>(inv0 + nn);
    const cd* psi0 = reinterpret_cast<const cd*>(psi0_d);
    if (gemm(ctx, psi0, psi0, Sig, n, 1, nn, nn, nn, 1, 0)) return -2;
    if (invert_one(ctx, inv0, psi0, n, inf)) return -2;
    if (gemm(ctx, reinterpret_cast<const cd*>(psi_d), inv0, T1, n, nf, nn, 0, nn, 0, 0)) return -2;
    const unsigned eb = (unsigned)std::min<size_t>((tot + 255) / 256, 8192);
    hipLaunchKernelGGL(spywil::granger_kernel, dim3(eb), dim3(256), 0, ctx->stream, reinterpret_cast<const cd*>(A_d), T1, Sig, nf, n,
                       reinterpret_cast<float*>(granger_d));
    SPY_HIP_CHECK(hipGetLastError());
    if (H_d) SPY_HIP_CHECK(hipMemcpyAsync(H_d, T1, tot * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
    if (Sigma_d) SPY_HIP_CHECK(hipMemcpyAsync(Sigma_d, Sig, nn * sizeof(cd), hipMemcpyDeviceToDevice, ctx->stream));
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}
