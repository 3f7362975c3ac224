// The compile-time schedules of the reference-precision tapered FFT (mtmfft_dec64_kernel.h) - one place: the launchers
// (mtmfft_dec64_launch.h), the plan's length table (mtmfft.hip) and the kernel emulator of the tests read it.
#pragma once
#include "mtmfft_dec64_kernel.h"

namespace spyfft {

//                       V   R1  R2  R3  G   SPLIT  XRES  HOIST
using D64_256   = CfgD64<16, 16, 1,  1,  16>;
using D64_512   = CfgD64<16, 16, 2,  1,  8>;
using D64_1024  = CfgD64<16, 16, 4,  1,  4>;
using D64_2048  = CfgD64<16, 16, 8,  1,  2>;
using D64_4096  = CfgD64<16, 16, 16, 1,  1>;
using D64_8192  = CfgD64<16, 16, 16, 2,  1, false, true, false>;
// 16384: 512 threads x 32 values, two split exchanges (a 16-value schedule needs 1024 threads = 128 registers per thread:
// measured 242 vs 204 us/trial for the complex spectra and 1040 vs 184 with the taper mean, profiles/r4_precision_probe.txt)
using D64_16384 = CfgD64<32, 32, 16, 1,  1, true, false, false>;   // (HOIST off: 532 vs 184 us/trial with the taper mean)
using D64_200   = CfgD64<10, 10, 2,  1,  8>;
using D64_500   = CfgD64<10, 10, 5,  1,  4>;
using D64_1000  = CfgD64<10, 10, 10, 1,  2>;
using D64_2000  = CfgD64<10, 10, 10, 2,  1>;
using D64_2500  = CfgD64<10, 10, 5,  5,  1>;
using D64_5000  = CfgD64<10, 10, 10, 5,  1>;
using D64_4000  = CfgD64<20, 20, 10, 1,  1>;
using D64_10000 = CfgD64<20, 20, 5,  5,  1, true>;
// 3 x (a schedule above): three sub-transforms side by side + one radix-3 combine (CfgD64::P)
//                       V   R1  R2  R3  G   SPLIT  XRES  HOIST P
using D64_600   = CfgD64<10, 10, 2,  1,  4, false, true, true, 3>;
using D64_1500  = CfgD64<10, 10, 5,  1,  2, false, true, true, 3>;
using D64_3000  = CfgD64<10, 10, 10, 1,  1, false, true, true, 3>;
using D64_6000  = CfgD64<10, 10, 10, 2,  1, false, true, true, 3>;
using D64_7500  = CfgD64<10, 10, 5,  5,  1, false, true, true, 3>;
using D64_768   = CfgD64<16, 16, 1,  1,  4, false, true, true, 3>;
using D64_1536  = CfgD64<16, 16, 2,  1,  2, false, true, true, 3>;
using D64_3072  = CfgD64<16, 16, 4,  1,  1, false, true, true, 3>;
using D64_6144  = CfgD64<16, 16, 8,  1,  1, false, true, true, 3>;
// HALF form: single channels, the real transform of 2 N samples through the schedule of N (trials of 12000 ... 20000 samples)
//                        V   R1  R2  R3  G   SPLIT  XRES  HOIST P  HALF
using D64H_12000 = CfgD64<10, 10, 10, 2,  1, false, true, true, 3, true>;
using D64H_12288 = CfgD64<16, 16, 8,  1,  1, false, true, true, 3, true>;
using D64H_15000 = CfgD64<10, 10, 5,  5,  1, false, true, true, 3, true>;
using D64H_16000 = CfgD64<20, 20, 20, 1,  1, false, true, true, 1, true>;
using D64H_16384 = CfgD64<16, 16, 16, 2,  1, false, true, false, 1, true>;
using D64H_20000 = CfgD64<20, 20, 5,  5,  1, true, true, true, 1, true>;
// window lengths of sliding-window analyses and the remaining multiples of 100 up to 8000 that factor into the radices
using D64_100   = CfgD64<10, 10, 1,  1,  16>;
using D64_400   = CfgD64<20, 20, 1,  1,  8>;      // (10 x 10 x 2 x 2, the float32 choice, measured 2.5 vs 2.2 us/trial here)
using D64_800   = CfgD64<20, 20, 2,  1,  4>;
using D64_1600  = CfgD64<20, 20, 4,  1,  2>;
using D64_3200  = CfgD64<20, 20, 4,  2,  1>;
using D64_8000  = CfgD64<20, 20, 20, 1,  1>;
using D64_300   = CfgD64<10, 10, 1,  1,  8, false, true, true, 3>;
using D64_1200  = CfgD64<20, 20, 1,  1,  2, false, true, true, 3>;
using D64_2400  = CfgD64<20, 20, 2,  1,  1, false, true, true, 3>;
using D64_4800  = CfgD64<20, 20, 4,  1,  1, false, true, true, 3>;

}  // namespace spyfft
