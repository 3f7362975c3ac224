// HBM ceilings the hand-over between the transform kernel (writes 29.4 MB of spectra per trial) and the cross-spectral
// kernel (reads them) should be read against: write-only, read-only and copy streams of 16 bytes per lane over 16 GiB.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/hbm_rw_probe tools/hbm_rw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) wr(float4* p, size_t n) {
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
__global__ void __launch_bounds__(256) wr64(float4* p, size_t n) {      // 64 contiguous bytes per group of 4 lanes, rows 2 KiB apart
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);   // (the transform kernel's pattern: lanes = different bins)
    const size_t lane4 = threadIdx.x & 3, row = (blockIdx.x * 256ull + threadIdx.x) >> 2;
    for (size_t r = row; r * 128 + 4 < n; r += (size_t)gridDim.x * 64)
        for (int q = 0; q < 32; ++q) p[r * 128 + q * 4 + lane4] = v;   // a 2-KiB row in 32 pieces of 64 bytes
}
__global__ void __launch_bounds__(256) rd(const float4* p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; s += v.x + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) cp(const float4* a, float4* b, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

int main() {
    const size_t bytes = 16ull << 30, n = bytes / 16;
    float4 *a, *b;
    float* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch, double traffic) {
        launch();
        hipEventRecord(e0);
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("%-34s %7.3f ms  %6.2f TB/s  %s\n", name, ms, traffic / ms * 1e-9, hipGetErrorString(hipGetLastError()));
    };
    for (int grid : {2048, 8192, 32768}) {
        printf("grid %d x 256 threads\n", grid);
        run("write only, 16 B per lane", [&] { wr<<<grid, 256>>>(a, n); }, (double)bytes);
        run("write only, 64-byte pieces of rows", [&] { wr64<<<grid, 256>>>(a, n); }, (double)bytes);
        run("read only, 16 B per lane", [&] { rd<<<grid, 256>>>(a, n, o); }, (double)bytes);
        run("copy (read + write counted)", [&] { cp<<<grid, 256>>>(a, b, n); }, 2.0 * bytes);
    }
    return 0;
}
