// Micro-benchmark: HBM write rate of a GEMM-epilogue-like store pattern on gfx950.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/store_pattern.hip -o scripts/ubench/store_pattern.bin
// A [M][N] fp32 matrix is written by workgroups of 256 threads, each owning a tile of BM rows x BN columns (float4 per
// thread and step, rows of the tile BN*4 bytes long at stride N*4), tiles in the order of the GEMM (column tile fastest).
// Prints the rate for several tile widths: BN = N is the fully contiguous case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NT>
__global__ __launch_bounds__(256) void store_tiles(float* __restrict__ C, float* __restrict__ C2, int M, int N, int BM, int BN,
                                                   int tiles_n, int two) {
    const int t = blockIdx.x;
    const int tn = t % tiles_n, tm = t / tiles_n;
    const int Q = BN / 4;
    const int n_it = BM * Q / 256;
    for (int it = 0; it < n_it; ++it) {
        const int e = threadIdx.x + it * 256;
        const int row = e / Q, q = e - row * Q;
        const long long off = (long long)(tm * BM + row) * N + tn * BN + q * 4;
        const float4 v = make_float4((float)e, 1.f, 2.f, 3.f);
        if (NT) {
            __builtin_nontemporal_store(v.x, C + off); __builtin_nontemporal_store(v.y, C + off + 1);
            __builtin_nontemporal_store(v.z, C + off + 2); __builtin_nontemporal_store(v.w, C + off + 3);
        } else {
            *reinterpret_cast<float4*>(C + off) = v;
            if (two) *reinterpret_cast<float4*>(C2 + off) = v;
        }
    }
}

int main() {
    const int M = 150528, N = 384;
    float *C, *C2;
    hipMalloc(&C, (size_t)M * N * 4);
    hipMalloc(&C2, (size_t)M * N * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { int BM, BN, two; };
    const Cfg cfgs[] = {{64, 128, 0}, {128, 128, 0}, {64, 384, 0}, {32, 384, 0}, {16, 384, 0}, {64, 128, 1}, {32, 384, 1}, {64, 384, 1}, {64, 96, 0}, {64, 192, 0}};
    for (const Cfg& c : cfgs) {
        const int tiles_n = N / c.BN, tiles_m = M / c.BM;
        auto run = [&]() { hipLaunchKernelGGL(store_tiles<0>, dim3(tiles_n * tiles_m), dim3(256), 0, 0, C, C2, M, N, c.BM, c.BN, tiles_n, c.two); };
        for (int i = 0; i < 3; ++i) run();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) run();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)M * N * 4 * (1 + c.two);
        printf("tile %3d x %3d, %d output(s): %7.1f us  %5.2f TB/s\n", c.BM, c.BN, 1 + c.two, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}
