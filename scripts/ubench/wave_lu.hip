// Microbenchmark (development aid): the 32 x 32 head factorisation of k_diag_lu2 (phase B, wave_lu32: one wave, lane = row, pivot rows
// broadcast with v_readlane) against an LDS-broadcast form, with the shader clock (s_memtime) and the constant 100 MHz clock
// (s_memrealtime) read around it: cycles per factorisation and the shader clock the lone wave actually runs at.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I superlu_dist_amd/csrc -I include -munsafe-fp-atomics scripts/ubench/wave_lu.hip -o scripts/ubench/wave_lu
#include "sluamd_kernels.hip"
#include <cstdio>
#include <vector>
using namespace sluamd;
namespace sluamd { void set_error(const std::string &) {} const std::string &get_error() { static std::string e; return e; } }

// LDS-broadcast variant: the pivot row goes through LDS (one masked ds_write per column by lane j, one uniform-address ds_read per column by
// everybody) instead of two v_readlane_b32 per column
__device__ __forceinline__ void wave_lu32_lds(double *P, int ld, int nb, double *rowbuf, double *s_rinv)
{
    const int lane = threadIdx.x & 63;
    double a[DB];
#pragma unroll
    for (int c = 0; c < DB; ++c) a[c] = (lane < nb && c < nb) ? P[c * ld + lane] : ((c == lane) ? 1.0 : 0.0);
#pragma unroll
    for (int j = 0; j < DB; ++j) {
        if (lane == j) {
#pragma unroll
            for (int c = j; c < DB; ++c) rowbuf[c] = a[c];
        }
        __builtin_amdgcn_wave_barrier();
        const double p = rowbuf[j];
        const double rinv = (p != 0.0) ? pivot_recip(p) : 1.0;
        if (lane == 0) s_rinv[j] = rinv;
        const bool below = lane > j;
        const double l = a[j] * rinv;
        if (below) a[j] = l;
        const double lm = below ? l : 0.0;
#pragma unroll
        for (int c = j + 1; c < DB; ++c) a[c] -= lm * rowbuf[c];
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int c = 0; c < DB; ++c) if (lane < nb && c < nb) P[c * ld + lane] = a[c];
}

template <int VAR>
__global__ __launch_bounds__(256) void k_bench(const double *mat, double *outm, long long *out, int reps, int *info)
{
    __shared__ double M0[DB * 33], P[DB * 33], rowbuf[DB], s_rinv[DB];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < DB * DB; i += 256) M0[(i >> 5) * 33 + (i & 31)] = mat[i];
    __syncthreads();
    long long c0 = 0, c1 = 0, w0 = 0, w1 = 0;
    if (wave == 0) {
        c0 = clock64(); w0 = wall_clock64();
        for (int r = 0; r < reps; ++r) {
            for (int i = tid; i < DB * 33; i += 64) P[i] = M0[i];
            if (VAR == 0) wave_lu32(P, 33, DB, 1, 0, 0.0, info, s_rinv);
            else wave_lu32_lds(P, 33, DB, rowbuf, s_rinv);
        }
        c1 = clock64(); w1 = wall_clock64();
    }
    __syncthreads();
    for (int i = tid; i < DB * DB; i += 256) outm[i] = P[(i >> 5) * 33 + (i & 31)];
    if (tid == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}

// throughput of v_readlane_b32 + v_fma_f64 with an SGPR operand, 32 independent chains
__global__ void k_readlane(double *o, long long *out, int reps)
{
    double a[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) a[c] = o[threadIdx.x + 64 * c];
    const double lm = o[threadIdx.x];
    long long c0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int c = 0; c < 32; ++c) { const double u = lane_bcast(a[c], (r + c) & 63); a[c] -= lm * u; }
    }
    long long c1 = clock64();
    double sacc = 0;
#pragma unroll
    for (int c = 0; c < 32; ++c) sacc += a[c];
    o[threadIdx.x] = sacc;
    if (threadIdx.x == 0) out[0] = c1 - c0;
}

int main()
{
    std::vector<double> h(DB * DB);
    for (int c = 0; c < DB; ++c) for (int r = 0; r < DB; ++r) h[c * DB + r] = (r == c) ? 40.0 + r : 1.0 / (1 + ((r * 7 + c * 13) % 11));
    double *d, *dm, *dr; long long *dout; int *dinfo;
    hipMalloc(&d, sizeof(double) * DB * DB); hipMalloc(&dm, sizeof(double) * DB * DB); hipMalloc(&dout, 16); hipMalloc(&dinfo, 16); hipMalloc(&dr, sizeof(double) * 64 * 33);
    hipMemcpy(d, h.data(), sizeof(double) * DB * DB, hipMemcpyHostToDevice);
    hipMemset(dinfo, 0, 16); hipMemset(dr, 0, sizeof(double) * 64 * 33);
    const int reps = 2000;
    std::vector<double> r0(DB * DB), r1(DB * DB);
    for (int var = 0; var < 2; ++var)
        for (int pass = 0; pass < 2; ++pass) {
            if (var == 0) hipLaunchKernelGGL(k_bench<0>, dim3(1), dim3(256), 0, 0, d, dm, dout, reps, dinfo);
            else hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(256), 0, 0, d, dm, dout, reps, dinfo);
            hipDeviceSynchronize();
            long long o[2]; hipMemcpy(o, dout, 16, hipMemcpyDeviceToHost);
            hipMemcpy((var ? r1 : r0).data(), dm, sizeof(double) * DB * DB, hipMemcpyDeviceToHost);
            printf("variant %d (%s) pass %d: %.0f shader cycles per 32 x 32 LU, %.2f us (100 MHz clock), shader clock %.0f MHz\n", var, var ? "LDS broadcast" : "v_readlane",
                   pass, (double) o[0] / reps, (double) o[1] / reps / 100.0, (double) o[0] / ((double) o[1] / 100.0));
        }
    double md = 0;
    for (int i = 0; i < DB * DB; ++i) md = fmax(md, fabs(r0[i] - r1[i]));
    printf("max |difference| between the two variants' factors: %.3e\n", md);
    hipLaunchKernelGGL(k_readlane, dim3(1), dim3(64), 0, 0, dr, dout, 1000);
    hipDeviceSynchronize();
    long long o[2]; hipMemcpy(o, dout, 16, hipMemcpyDeviceToHost);
    printf("v_readlane x2 + v_fma_f64 (SGPR operand): %.1f cycles per column update (lone wave)\n", (double) o[0] / 1000 / 32);
    return 0;
}
