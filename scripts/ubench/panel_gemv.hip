// Microbenchmark (development aid): y -= L x for ONE column-major panel (R rows, 256 columns, leading dimension R) -- the forward
// update of a single separator supernode in the triangular solve -- under different work shapes.  Each launch is timed alone
// (events around 20 back-to-back launches on distinct panels, so nothing is warm in L2 / MALL).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// V0: 1024 threads = (64 rows) x (16 column slices), one batch of 16 loads per thread (the library's kernel)
template <bool NT_LOAD>
__global__ __launch_bounds__(1024) void k_v0(const double *__restrict__ L, int R, int ns, const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ double xk[256];
    __shared__ double s_red[16][65];
    const int tid = threadIdx.x;
    if (tid < ns) xk[tid] = x[tid];
    __syncthreads();
    const int r = tid & 63, part = tid >> 6;
    const int row = blockIdx.x * 64 + r;
    const bool ok = row < R;
    const double *p = L + row;
    double acc[4] = {0, 0, 0, 0};
    if (ok) {
        const int ka = part * 16;
        double lv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) lv[u] = NT_LOAD ? __builtin_nontemporal_load(p + (size_t) (ka + u) * R) : p[(size_t) (ka + u) * R];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] += lv[u] * xk[ka + u];
    }
    s_red[part][r] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (part == 0 && ok) {
        double a = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += s_red[q][r];
        unsafeAtomicAdd(y + row, -a);
    }
}

// V1: 256 threads = (64 rows) x (4 slices), 4 batches of 16 loads per thread
__global__ __launch_bounds__(256) void k_v1(const double *__restrict__ L, int R, int ns, const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ double xk[256];
    __shared__ double s_red[4][65];
    const int tid = threadIdx.x;
    if (tid < ns) xk[tid] = x[tid];
    __syncthreads();
    const int r = tid & 63, part = tid >> 6;
    const int row = blockIdx.x * 64 + r;
    const bool ok = row < R;
    const double *p = L + row;
    double acc[4] = {0, 0, 0, 0};
    if (ok)
        for (int b = 0; b < 4; ++b) {
            const int ka = part * 64 + b * 16;
            double lv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) lv[u] = __builtin_nontemporal_load(p + (size_t) (ka + u) * R);
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 3] += lv[u] * xk[ka + u];
        }
    s_red[part][r] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (part == 0 && ok) unsafeAtomicAdd(y + row, -((s_red[0][r] + s_red[1][r]) + (s_red[2][r] + s_red[3][r])));
}

// V2: 256 threads, 32-row strips: (32 rows) x (8 slices), 2 batches of 16 -- twice the workgroups of V1, half-wave segments
__global__ __launch_bounds__(256) void k_v2(const double *__restrict__ L, int R, int ns, const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ double xk[256];
    __shared__ double s_red[8][33];
    const int tid = threadIdx.x;
    if (tid < ns) xk[tid] = x[tid];
    __syncthreads();
    const int r = tid & 31, part = tid >> 5;
    const int row = blockIdx.x * 32 + r;
    const bool ok = row < R;
    const double *p = L + row;
    double acc[4] = {0, 0, 0, 0};
    if (ok)
        for (int b = 0; b < 2; ++b) {
            const int ka = part * 32 + b * 16;
            double lv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) lv[u] = __builtin_nontemporal_load(p + (size_t) (ka + u) * R);
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 3] += lv[u] * xk[ka + u];
        }
    s_red[part][r] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (part == 0 && ok) {
        double a = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) a += s_red[q][r];
        unsafeAtomicAdd(y + row, -a);
    }
}

// V3: 256 threads, 128-row strips with 16-byte loads: lane owns 2 consecutive rows, (64 lanes x 2 rows) x (4 slices), 4 batches of 16
__global__ __launch_bounds__(256) void k_v3(const double *__restrict__ L, int R, int ns, const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ double xk[256];
    __shared__ double2 s_red[4][65];
    const int tid = threadIdx.x;
    if (tid < ns) xk[tid] = x[tid];
    __syncthreads();
    const int r = tid & 63, part = tid >> 6;
    const int row = blockIdx.x * 128 + 2 * r;
    const bool ok = row + 1 < R;     // R even in this benchmark
    const double *p = L + row;
    double a0 = 0, a1 = 0;
    if (ok)
        for (int b = 0; b < 4; ++b) {
            const int ka = part * 64 + b * 16;
            double2 lv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) lv[u] = *(const double2 *) (p + (size_t) (ka + u) * R);
#pragma unroll
            for (int u = 0; u < 16; ++u) { a0 += lv[u].x * xk[ka + u]; a1 += lv[u].y * xk[ka + u]; }
        }
    s_red[part][r] = make_double2(a0, a1);
    __syncthreads();
    if (part == 0 && ok) {
        double b0 = 0, b1 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { b0 += s_red[q][r].x; b1 += s_red[q][r].y; }
        unsafeAtomicAdd(y + row, -b0); unsafeAtomicAdd(y + row + 1, -b1);
    }
}

// V4: 1024 threads, 64-row strips over HALF of the columns (two workgroups per strip): 8 loads per thread
__global__ __launch_bounds__(1024) void k_v4(const double *__restrict__ L, int R, int ns, const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ double xk[256];
    __shared__ double s_red[16][65];
    const int tid = threadIdx.x;
    if (tid < ns) xk[tid] = x[tid];
    __syncthreads();
    const int strip = blockIdx.x >> 1, half = blockIdx.x & 1;
    const int r = tid & 63, part = tid >> 6;
    const int row = strip * 64 + r;
    const bool ok = row < R;
    const double *p = L + row;
    double acc[4] = {0, 0, 0, 0};
    if (ok) {
        const int ka = half * 128 + part * 8;
        double lv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) lv[u] = __builtin_nontemporal_load(p + (size_t) (ka + u) * R);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u & 3] += lv[u] * xk[ka + u];
    }
    s_red[part][r] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (part == 0 && ok) {
        double a = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += s_red[q][r];
        unsafeAtomicAdd(y + row, -a);
    }
}

// empty kernel of the same grid: the launch / dispatch floor
__global__ __launch_bounds__(1024) void k_empty(double *y) { if (threadIdx.x == 2000) y[0] = 1; }

int main()
{
    const int ns = 256, NP = 24;
    for (int R : {2048, 5120, 10000, 10240, 20000}) {
        const size_t psz = (size_t) R * ns;
        double *L, *x, *y;
        hipMalloc(&L, psz * NP * 8); hipMalloc(&x, 256 * 8); hipMalloc(&y, (size_t) R * 8);
        hipMemset(L, 0, psz * NP * 8); hipMemset(x, 0, 256 * 8); hipMemset(y, 0, (size_t) R * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto timeit = [&](const char *name, auto launch) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int p = 0; p < NP; ++p) launch(L + psz * p);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("R %6d  %-44s %7.2f us per launch  %6.2f TB/s\n", R, name, best * 1e3 / NP, (double) psz * 8 / (best / NP * 1e-3) / 1e12);
        };
        timeit("empty, 1024 x ceil(R/64)", [&](double *) { hipLaunchKernelGGL(k_empty, dim3((R + 63) / 64), dim3(1024), 0, 0, y); });
        timeit("v0 1024thr 64rows 16 loads nontemporal", [&](double *P) { hipLaunchKernelGGL(k_v0<true>, dim3((R + 63) / 64), dim3(1024), 0, 0, P, R, ns, x, y); });
        timeit("v0 1024thr 64rows 16 loads plain", [&](double *P) { hipLaunchKernelGGL(k_v0<false>, dim3((R + 63) / 64), dim3(1024), 0, 0, P, R, ns, x, y); });
        timeit("v1 256thr 64rows 4x16 loads", [&](double *P) { hipLaunchKernelGGL(k_v1, dim3((R + 63) / 64), dim3(256), 0, 0, P, R, ns, x, y); });
        timeit("v2 256thr 32rows 2x16 loads", [&](double *P) { hipLaunchKernelGGL(k_v2, dim3((R + 31) / 32), dim3(256), 0, 0, P, R, ns, x, y); });
        timeit("v3 256thr 128rows 4x16 16-byte loads", [&](double *P) { hipLaunchKernelGGL(k_v3, dim3((R + 127) / 128), dim3(256), 0, 0, P, R, ns, x, y); });
        timeit("v4 1024thr 64rows x half the columns 8 loads", [&](double *P) { hipLaunchKernelGGL(k_v4, dim3(2 * ((R + 63) / 64)), dim3(1024), 0, 0, P, R, ns, x, y); });
        hipFree(L); hipFree(x); hipFree(y);
    }
    return 0;
}
