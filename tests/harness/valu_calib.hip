// valu_calib.hip -- calibration kernels for the VALU counters and for what an execution mask costs.
//
// Each kernel runs ITER iterations of 32 independent register-to-register v_fma_f32 (or
// v_pk_fma_f32) instructions per wave, with only the lanes of a given 64-bit mask enabled.  The
// host times every (mask, waves per SIMD) combination and prints wave-instructions per SIMD per
// cycle, so that
//   - the issue ceiling of plain and packed f32 FMAs is known from a kernel of known VALU density,
//   - it is known whether disabled lanes cost anything: a wave64 VALU instruction is executed in
//     passes of 16 lanes, and if a pass whose lanes are all disabled were skipped, a mask that
//     keeps its active lanes in few 16-lane groups would run faster than a scattered one.
// Under rocprofv3 --pmc the same run calibrates SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES
// and SQ_THREAD_CYCLES_VALU against these known instruction counts.
//
// Test infrastructure; build: hipcc -O2 --offload-arch=gfx950 -o ../_build/valu_calib valu_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                            \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int kFmaPerIter = 32;

template <bool PACKED>
__global__ __launch_bounds__(256) void fma_kernel(float *out, unsigned long long mask, int iters, unsigned long long *cycles)
{
    const int lane = threadIdx.x & 63;
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999f, c = 1e-3f;
    float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    typedef float float2_ __attribute__((ext_vector_type(2)));
    float2_ p0 = {a, a + 1}, p1 = {a + 2, a + 3}, p2 = {a + 4, a + 5}, p3 = {a + 6, a + 7}, pb = {b, b}, pc = {c, c};
    unsigned long long t0 = 0, t1 = 0;
    if ((mask >> lane) & 1ull) {
        t0 = clock64();
        for (int i = 0; i < iters; i++) {
            if (!PACKED) {
#pragma unroll
                for (int k = 0; k < kFmaPerIter / 8; k++) {
                    asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\t"
                                 "v_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"
                                 "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                                 : "v"(b), "v"(c));
                }
            } else {
#pragma unroll
                for (int k = 0; k < kFmaPerIter / 4; k++) {
                    asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\t"
                                 "v_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                                 : "v"(pb), "v"(pc));
                }
            }
        }
        t1 = clock64();
    }
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (s == 123.456f) out[0] = s; // never true: keeps the chains alive
    if (cycles && lane == (int)__builtin_ctzll(mask) && blockIdx.x == 0 && threadIdx.x < 64) cycles[0] = t1 - t0;
}

int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    // optional filter, for runs under rocprofv3 --pmc: kind (fma | pk), mask number, waves per SIMD
    int only_packed = argc > 2 ? (argv[2][0] == 'p' ? 1 : 0) : -1;
    int only_mask = argc > 3 ? atoi(argv[3]) : -1;
    int only_wps = argc > 4 ? atoi(argv[4]) : -1;
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    int cus = prop.multiProcessorCount;
    double clock_hz = prop.clockRate * 1e3; // kHz -> Hz (the peak engine clock)
    printf("device %s, %d CUs, peak clock %.0f MHz, %d iterations of %d FMAs\n", prop.name, cus, clock_hz / 1e6, iters, kFmaPerIter);
    float *out;
    unsigned long long *cyc;
    CHECK(hipMalloc((void **)&out, 64));
    CHECK(hipMalloc((void **)&cyc, 64));
    struct M { const char *name; unsigned long long mask; };
    unsigned long long every4 = 0, rnd = 0x9e3779b97f4a7c15ull & 0xffff0000ffff0000ull;
    for (int i = 0; i < 64; i += 4) every4 |= 1ull << i;
    std::vector<M> masks = {
        {"all 64 lanes", ~0ull},
        {"lanes 0-31 (two 16-lane groups)", 0xffffffffull},
        {"lanes 0-15 (one 16-lane group)", 0xffffull},
        {"even lanes (32 lanes, all four groups)", 0x5555555555555555ull},
        {"every 4th lane (16 lanes, all four groups)", every4},
        {"lanes 0-15 and 32-47 (32 lanes, two groups)", 0x0000ffff0000ffffull},
        {"one lane", 1ull},
    };
    (void)rnd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-8s %-46s %5s %10s %12s %14s %14s\n", "kind", "mask", "w/SIMD", "ms", "wave-clk/FMA", "FMA/SIMD/clk", "lane-TFLOP/s");
    for (int packed = 0; packed < 2; packed++) {
        if (only_packed >= 0 && packed != only_packed) continue;
        for (size_t mi = 0; mi < masks.size(); mi++) {
            if (only_mask >= 0 && (int)mi != only_mask) continue;
            const M &m = masks[mi];
            for (int wps : {1, 2, 4, 8}) {
                if (only_wps >= 0 && wps != only_wps) continue;
                int blocks = cus * wps; // 256 threads = 4 waves = one wave per SIMD per block
                // warm up, then time
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(e0, 0));
                    if (packed) hipLaunchKernelGGL(fma_kernel<true>, dim3(blocks), dim3(256), 0, 0, out, m.mask, iters, cyc);
                    else hipLaunchKernelGGL(fma_kernel<false>, dim3(blocks), dim3(256), 0, 0, out, m.mask, iters, cyc);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                }
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                unsigned long long wc = 0;
                CHECK(hipMemcpy(&wc, cyc, 8, hipMemcpyDeviceToHost));
                double n_inst = (double)iters * kFmaPerIter;                 // per wave
                double per_simd = n_inst * wps / (ms * 1e-3 * clock_hz);     // wave-instructions per SIMD per peak clock
                int lanes = __builtin_popcountll(m.mask);
                double tflops = n_inst * wps * 4.0 * cus * lanes * (packed ? 4.0 : 2.0) / (ms * 1e-3) / 1e12;
                printf("%-8s %-46s %5d %10.3f %12.3f %14.3f %14.2f\n", packed ? "pk_fma" : "fma", m.name, wps, ms,
                       (double)wc / n_inst, per_simd, tflops);
            }
        }
    }
    return 0;
}
