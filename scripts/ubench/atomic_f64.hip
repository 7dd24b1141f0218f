// Microbenchmark (development aid): throughput of the k_schur scatter pattern -- every workgroup (512 threads) updates a 128 x 128
// fp64 tile (column stride ld) with (a) global_atomic_add_f64, (b) plain load + store -- cold (tiles spread over 4 GB) or warm (L2).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k(double *base, long tile_stride, int ld, int ntiles_mod, int reps)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rm0 = (wave & 3) * 32, cn0 = (wave >> 2) * 64;
    for (int it = 0; it < reps; ++it) {
        double *dst = base + (long) ((blockIdx.x + it * gridDim.x) % ntiles_mod) * tile_stride;
        if (MODE == 0) {
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double *dcol = dst + (long) (cn0 + 16 * ci + (lane >> 4) + 4 * r) * ld;
#pragma unroll
                    for (int ri = 0; ri < 2; ++ri) unsafeAtomicAdd(dcol + rm0 + 16 * ri + (lane & 15), 1.0);
                }
        } else {
            double v[32];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double *dcol = dst + (long) (cn0 + 16 * ci + (lane >> 4) + 4 * r) * ld;
#pragma unroll
                    for (int ri = 0; ri < 2; ++ri) v[(ci * 4 + r) * 2 + ri] = dcol[rm0 + 16 * ri + (lane & 15)];
                }
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double *dcol = dst + (long) (cn0 + 16 * ci + (lane >> 4) + 4 * r) * ld;
#pragma unroll
                    for (int ri = 0; ri < 2; ++ri) dcol[rm0 + 16 * ri + (lane & 15)] = v[(ci * 4 + r) * 2 + ri] + 1.0;
                }
        }
    }
}
template <int MODE> void run(const char *name, double *buf, long tile_stride, int ld, int ntiles_mod, int grid, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(512), 0, 0, buf, tile_stride, ld, ntiles_mod, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double elems = (double) grid * reps * 16384.0;
        if (w) printf("%-46s %8.3f ms  %7.1f G elem/s  %6.2f us per tile per CU-slot (512 slots)  %.2f TB/s RMW traffic\n", name, ms, elems / ms / 1e6,
                      ms * 1e3 / ((double) grid * reps / 512.0), elems * 16 / ms / 1e9);
    }
}
int main()
{
    const long total = 4L << 30;   // 4 GB of doubles = 32 GB? no: bytes
    double *buf; hipMalloc(&buf, (size_t) total); hipMemset(buf, 0, (size_t) total);
    const int ld = 3000;           // a destination panel with 3000 rows
    const long tile_stride_cold = 128L * ld + 1024;   // distinct tiles
    const int ncold = (int) (total / 8 / tile_stride_cold) - 1;
    run<0>("atomic, cold destinations", buf, tile_stride_cold, ld, ncold, 8192, 4);
    run<1>("load+store, cold destinations", buf, tile_stride_cold, ld, ncold, 8192, 4);
    run<0>("atomic, 64 tiles re-used (L2/MALL warm)", buf, tile_stride_cold, ld, 64, 8192, 4);
    run<1>("load+store, 64 tiles re-used", buf, tile_stride_cold, ld, 64, 8192, 4);
    run<0>("atomic, 1024 tiles re-used (MALL)", buf, tile_stride_cold, ld, 1024, 8192, 4);
    run<1>("load+store, 1024 tiles re-used (MALL)", buf, tile_stride_cold, ld, 1024, 8192, 4);
    return 0;
}
