// Microbenchmark: what a CU mask on a HIP stream (hipExtStreamCreateWithCUMask) does on MI355X -- which (XCC, SE, CU) the workgroups of a launch land on for
// a few masks, and how a one-wave latency-bound kernel (a dependent chain of fp64 FMAs, the shape of the panel-chain LU kernels) runs beside an MFMA-saturating
// kernel with and without disjoint masks.
//   hipcc --offload-arch=gfx950 -O3 -o cumask cumask.hip && ./cumask
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
#include <map>
#include <chrono>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_where(uint32_t *out)
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    // stay a little so that the launch spreads over the allowed CUs
    double a = threadIdx.x;
    for (int i = 0; i < 2000; ++i) a = a * 1.0000001 + 1e-9;
    if (a == 12345.0) out[0] = 0;
}

typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma_hog(double *out, int iters)
{
    d4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
    }
    if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 1.2345) out[0] = 1;
}

__global__ __launch_bounds__(64) void k_chain(double *out, int iters)
{   // one wave: dependent fp64 FMAs with a cross-lane broadcast per step -- the inner loop of the one-wave LU kernels
    double a = threadIdx.x * 1e-3 + 1.0;
    for (int i = 0; i < iters; ++i) {
        const double p = __shfl(a, i & 63);
        a = a * 0.999999 + p * 1e-9;
    }
    out[threadIdx.x] = a;
}

static int run_where(hipStream_t s, const char *what, uint32_t *d, int nwg)
{
    std::vector<uint32_t> h(2 * nwg);
    hipLaunchKernelGGL(k_where, dim3(nwg), dim3(256), 0, s, d);
    CHK(hipStreamSynchronize(s));
    CHK(hipMemcpy(h.data(), d, sizeof(uint32_t) * 2 * nwg, hipMemcpyDeviceToHost));
    std::map<int, std::set<int>> per_xcc;
    for (int i = 0; i < nwg; ++i) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
    }
    int tot = 0;
    printf("%-44s", what);
    for (auto &kv : per_xcc) { printf(" xcc%d:%zu", kv.first, kv.second.size()); tot += (int) kv.second.size(); }
    printf("  -> %d distinct CUs\n", tot);
    return 0;
}

int main()
{
    hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    printf("device %s, %d CUs\n", pr.gcnArchName, ncu);
    const int words = (ncu + 31) / 32;
    uint32_t *d; CHK(hipMalloc(&d, sizeof(uint32_t) * 2 * 65536));
    double *dd; CHK(hipMalloc(&dd, 4096));
    hipStream_t s_all; CHK(hipStreamCreateWithFlags(&s_all, hipStreamNonBlocking));
    run_where(s_all, "no mask", d, 16384);
    auto mk = [&](auto pred, hipStream_t *s) { std::vector<uint32_t> m(words, 0); for (int i = 0; i < ncu; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); return hipExtStreamCreateWithCUMask(s, words, m.data()); };
    hipStream_t s_lo8, s_lo32, s_ev, s_hi, s_first_of8;
    CHK(mk([](int i) { return i < 8; }, &s_lo8));             run_where(s_lo8, "bits 0..7", d, 16384);
    CHK(mk([](int i) { return i < 32; }, &s_lo32));           run_where(s_lo32, "bits 0..31", d, 16384);
    CHK(mk([](int i) { return i % 2 == 0; }, &s_ev));         run_where(s_ev, "even bits", d, 16384);
    CHK(mk([](int i) { return i >= 8; }, &s_hi));             run_where(s_hi, "bits 8..", d, 16384);
    CHK(mk([](int i) { return i % 32 == 0; }, &s_first_of8)); run_where(s_first_of8, "bits 0, 32, 64, ...", d, 16384);

    // a latency-bound wave beside an MFMA-saturating launch
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto time_chain = [&](hipStream_t sc, hipStream_t sh, bool hog, const char *what) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            if (hog) hipLaunchKernelGGL(k_mfma_hog, dim3(ncu * 8), dim3(256), 0, sh, dd + 256, 40000);
            // let the hog occupy the device before the chain starts
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, sc, dd, 10);
            hipStreamSynchronize(sc);
            hipEventRecord(e0, sc);
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, sc, dd, 20000);
            hipEventRecord(e1, sc);
            hipStreamSynchronize(sc);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            hipDeviceSynchronize();
        }
        printf("%-64s %8.1f us\n", what, best * 1e3f);
    };
    time_chain(s_all, s_all, false, "chain alone");
    hipStream_t s_all2; CHK(hipStreamCreateWithFlags(&s_all2, hipStreamNonBlocking));
    time_chain(s_all, s_all2, true, "chain beside MFMA hog, no masks");
    time_chain(s_lo8, s_all2, true, "chain on bits 0..7, hog unmasked");
    time_chain(s_lo8, s_hi, true, "chain on bits 0..7, hog on bits 8..");
    {   // how long the hog takes with and without the reserved CUs
        for (auto pr2 : {std::make_pair(s_all2, "hog unmasked"), std::make_pair(s_hi, "hog on bits 8..")}) {
            hipEventRecord(e0, pr2.first);
            hipLaunchKernelGGL(k_mfma_hog, dim3(ncu * 8), dim3(256), 0, pr2.first, dd + 256, 40000);
            hipEventRecord(e1, pr2.first);
            hipStreamSynchronize(pr2.first);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("%-64s %8.1f us\n", pr2.second, ms * 1e3f);
        }
    }
    {   // cost of a dependent hop between two streams (event record on one, wait on the other, a one-wave kernel each side): normal against CU-masked streams
        hipStream_t n1, n2; CHK(hipStreamCreateWithFlags(&n1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&n2, hipStreamNonBlocking));
        std::vector<hipEvent_t> ev(2048);
        for (size_t i = 0; i < ev.size(); ++i) CHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        auto pingpong = [&](hipStream_t a, hipStream_t b, const char *what) {
            const int hops = 500;
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            hipEventRecord(e0, a);
            for (int i = 0; i < hops; ++i) {
                hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, a, dd, 1);
                hipEventRecord(ev[2 * i], a); hipStreamWaitEvent(b, ev[2 * i], 0);
                hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, b, dd + 64, 1);
                hipEventRecord(ev[2 * i + 1], b); hipStreamWaitEvent(a, ev[2 * i + 1], 0);
            }
            hipEventRecord(e1, a);
            const auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(a);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("%-64s %6.1f us per hop on the device, %6.1f us of host time per hop\n", what, ms * 1e3f / (2 * hops), std::chrono::duration<double, std::micro>(t1 - t0).count() / (2 * hops));
        };
        pingpong(n1, n2, "ping-pong between two ordinary streams");
        pingpong(n1, n1, "the same launches on ONE ordinary stream");
        pingpong(s_lo8, s_hi, "ping-pong between two CU-masked streams");
        pingpong(n1, s_lo8, "ping-pong ordinary <-> CU-masked");
        pingpong(s_hi, s_hi, "the same launches on ONE CU-masked stream");
    }
    return 0;
}
