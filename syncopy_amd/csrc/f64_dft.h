// Complex128 in-register DFTs of the radices the compile-time schedules of mtmfft_dec64_kernel.h use:
// 2 / 4 / 8 / 16 (wilson_plus_kernel.h), 3 / 5 (real-arithmetic butterflies) and the composites 6 / 10 / 12 / 20 / 32
// (Cooley-Tukey inside one thread's registers, internal twiddles as compile-time constants).
#pragma once
#include "cd_math.h"
#include "wilson_plus_kernel.h"      // p_dft4, p_dft16, p_dftR<2|4|8>
#include "mtmfft_mixed.h"            // mx_cos_turn / mx_sin_turn: constexpr roots of unity in double precision

namespace spywil {

template <int R>
__device__ __forceinline__ void d_dft(cd (&t)[R]);

__device__ __forceinline__ void d_dft3(cd (&t)[3]) {
    constexpr double s3 = 0.86602540378443864676;
    const cd s = cadd(t[1], t[2]), d = csub(t[1], t[2]);
    const cd m = make_double2(t[0].x - 0.5 * s.x, t[0].y - 0.5 * s.y);
    const cd e = make_double2(d.y * s3, -d.x * s3);                  // -i sin(2 pi / 3) d
    t[0] = cadd(t[0], s);
    t[1] = cadd(m, e);
    t[2] = csub(m, e);
}

__device__ __forceinline__ void d_dft5(cd (&t)[5]) {
    constexpr double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
    constexpr double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
    const cd a1 = cadd(t[1], t[4]), a2 = cadd(t[2], t[3]), b1 = csub(t[1], t[4]), b2 = csub(t[2], t[3]);
    const cd m1 = make_double2(t[0].x + a1.x * c1 + a2.x * c2, t[0].y + a1.y * c1 + a2.y * c2);
    const cd m2 = make_double2(t[0].x + a1.x * c2 + a2.x * c1, t[0].y + a1.y * c2 + a2.y * c1);
    const cd n1 = make_double2(b1.x * s1 + b2.x * s2, b1.y * s1 + b2.y * s2);
    const cd n2 = make_double2(b1.x * s2 - b2.x * s1, b1.y * s2 - b2.y * s1);
    t[0] = cadd(t[0], cadd(a1, a2));
    t[1] = make_double2(m1.x + n1.y, m1.y - n1.x);                    // m1 - i n1
    t[4] = make_double2(m1.x - n1.y, m1.y + n1.x);
    t[2] = make_double2(m2.x + n2.y, m2.y - n2.x);
    t[3] = make_double2(m2.x - n2.y, m2.y + n2.x);
}

// R = P Q: n = Q n1 + n2, k = k1 + P k2
template <int P, int Q>
__device__ __forceinline__ void d_dft_pq(cd (&t)[P * Q]) {
    constexpr int R = P * Q;
    cd y[Q][P];
#pragma unroll
    for (int n2 = 0; n2 < Q; ++n2) {
        cd u[P];
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) u[n1] = t[Q * n1 + n2];
        d_dft<P>(u);
#pragma unroll
        for (int k1 = 0; k1 < P; ++k1) {
            const double c = spyfft::mx_cos_turn(n2 * k1, R), s = spyfft::mx_sin_turn(n2 * k1, R);
            y[n2][k1] = (n2 * k1 == 0) ? u[k1] : cmul(u[k1], make_double2(c, -s));
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < P; ++k1) {
        cd u[Q];
#pragma unroll
        for (int n2 = 0; n2 < Q; ++n2) u[n2] = y[n2][k1];
        d_dft<Q>(u);
#pragma unroll
        for (int k2 = 0; k2 < Q; ++k2) t[k1 + P * k2] = u[k2];
    }
}

template <int R>
__device__ __forceinline__ void d_dft(cd (&t)[R]) {
    if constexpr (R == 2 || R == 4 || R == 8) p_dftR<R>(t);
    else if constexpr (R == 3) d_dft3(t);
    else if constexpr (R == 5) d_dft5(t);
    else if constexpr (R == 6) d_dft_pq<2, 3>(t);
    else if constexpr (R == 10) d_dft_pq<2, 5>(t);
    else if constexpr (R == 12) d_dft_pq<4, 3>(t);
    else if constexpr (R == 15) d_dft_pq<3, 5>(t);
    else if constexpr (R == 16) p_dft16(t);
    else if constexpr (R == 20) d_dft_pq<4, 5>(t);
    else if constexpr (R == 30) d_dft_pq<5, 6>(t);
    else if constexpr (R == 32) d_dft_pq<2, 16>(t);
    else static_assert(R == 2, "radix not built");
}

}  // namespace spywil
