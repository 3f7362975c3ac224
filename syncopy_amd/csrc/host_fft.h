// Small fp64 host FFT (iterative radix-2) used once per plan to build filter spectra
// (Bluestein chirp filter, Morlet kernel spectra).
#pragma once
#include <cmath>
#include <utility>
#include <vector>

namespace spy {
inline void fft_host(std::vector<double>& re, std::vector<double>& im) {
    const double PI = 3.14159265358979323846264338327950288;
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            std::swap(re[i], re[j]);
            std::swap(im[i], im[j]);
        }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * PI / (double)len;
        for (size_t k = 0; k < len / 2; ++k) {
            const double wr = std::cos(ang * k), wi = std::sin(ang * k);
            for (size_t i = 0; i < n; i += len) {
                const size_t a = i + k, b = i + k + len / 2;
                const double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr;
                im[b] = im[a] - ti;
                re[a] += tr;
                im[a] += ti;
            }
        }
    }
}
}  // namespace spy
