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

// v_fma_mix_f32 reading its first operand as the low or high half of a register (how half-precision
// node planes would be consumed without conversion instructions), the rest as f32.
__global__ __launch_bounds__(256) void fma_mix_kernel(float *out, unsigned long long mask, int iters)
{
    const int lane = threadIdx.x & 63;
    float b = 0.999f, c = 1e-3f;
    unsigned h0 = 0x3c003c01u + threadIdx.x, h1 = 0x3c023c03u, h2 = 0x3c043c05u, h3 = 0x3c063c07u;
    float r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < kFmaPerIter / 8; k++) {
                asm volatile("v_fma_mix_f32 %0, %8, %12, %13 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %8, %12, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                             "v_fma_mix_f32 %2, %9, %12, %13 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %9, %12, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                             "v_fma_mix_f32 %4, %10, %12, %13 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, %10, %12, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                             "v_fma_mix_f32 %6, %11, %12, %13 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %7, %11, %12, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                             : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)
                             : "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(b), "v"(c));
            }
        }
    }
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if (s == 123.456f) out[0] = s;
}

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

// Vector-memory counterpart: ITER iterations of 8 independent 16-byte loads per lane from a 64 KB
// table that stays in the L1 (each lane reads its own 16 bytes of a 1 KB row, rows chosen per load),
// under the same execution masks: whether a load instruction with half of its lanes disabled takes
// half the time in the address and data units.
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void load_kernel(const v4f *table, float *out, unsigned long long mask, int iters, int spread)
{
    const int lane = threadIdx.x & 63;
    v4f acc = {0, 0, 0, 0};
    if ((mask >> lane) & 1ull) {
        // spread 0: lane-contiguous rows (1 KB per load instruction); 1: every lane in a row of its own (divergent, as node fetches are)
        unsigned idx = spread ? (unsigned)(lane * 61 + threadIdx.x / 64 * 7) : (unsigned)lane;
        for (int i = 0; i < iters; i++) {
            v4f v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                unsigned row = (unsigned)(i * 8 + k) & 63u;
                unsigned at = spread ? ((idx + row * 67u) & 4095u) : (row * 64u + idx);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[k]) : "v"(table + at));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("" ::"v"(v[k]));
            acc.x += v[0].x;
        }
    }
    if (acc.x == 123.456f) out[0] = acc.x;
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 'm') { // "m [iters]": v_fma_mix_f32 issue rate
        int iters = argc > 2 ? atoi(argv[2]) : 20000;
        CHECK(hipSetDevice(0));
        hipDeviceProp_t prop;
        CHECK(hipGetDeviceProperties(&prop, 0));
        int cus = prop.multiProcessorCount;
        double clock_hz = prop.clockRate * 1e3;
        float *out;
        CHECK(hipMalloc((void **)&out, 64));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        for (int wps : {1, 2, 4, 6, 8}) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(fma_mix_kernel, dim3(cus * wps), dim3(256), 0, 0, out, ~0ull, iters);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("fma_mix  all 64 lanes  %d waves/SIMD  %.3f ms  %.3f wave-instructions/SIMD/clk\n", wps, ms,
                   (double)iters * kFmaPerIter * wps / (ms * 1e-3 * clock_hz));
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'l') { // "l [iters]": the load kernel
        int iters = argc > 2 ? atoi(argv[2]) : 4000;
        CHECK(hipSetDevice(0));
        hipDeviceProp_t prop;
        CHECK(hipGetDeviceProperties(&prop, 0));
        int cus = prop.multiProcessorCount;
        double clock_hz = prop.clockRate * 1e3;
        v4f *table;
        float *out;
        CHECK(hipMalloc((void **)&table, 4096 * sizeof(v4f)));
        CHECK(hipMemset(table, 0, 4096 * sizeof(v4f)));
        CHECK(hipMalloc((void **)&out, 64));
        unsigned long long every4 = 0;
        for (int i = 0; i < 64; i += 4) every4 |= 1ull << i;
        struct M { const char *name; unsigned long long mask; };
        std::vector<M> masks = {{"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"even lanes", 0x5555555555555555ull},
                                {"lanes 0-15", 0xffffull}, {"every 4th lane", every4}, {"one lane", 1ull}};
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        printf("%-10s %-18s %6s %10s %22s\n", "addresses", "mask", "w/SIMD", "ms", "CU clocks per load inst");
        for (int spread = 0; spread < 2; spread++)
            for (const M &m : masks)
                for (int wps : {2, 6}) {
                    int blocks = cus * wps;
                    float ms = 0;
                    for (int rep = 0; rep < 2; rep++) {
                        CHECK(hipEventRecord(e0, 0));
                        hipLaunchKernelGGL(load_kernel, dim3(blocks), dim3(256), 0, 0, table, out, m.mask, iters, spread);
                        CHECK(hipEventRecord(e1, 0));
                        CHECK(hipEventSynchronize(e1));
                        CHECK(hipEventElapsedTime(&ms, e0, e1));
                    }
                    double insts_per_cu = (double)iters * 8 * wps * 4; // wave-level load instructions per CU
                    printf("%-10s %-18s %6d %10.3f %22.2f\n", spread ? "divergent" : "contiguous", m.name, wps, ms,
                           ms * 1e-3 * clock_hz / insts_per_cu);
                }
        return 0;
    }
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
