// Issue cost of the VALU / LDS instructions the masking kernel is made of, in shader-clock cycles per wave64
// instruction on one SIMD: one wave (or W waves of one workgroup on the same SIMD... W waves per SIMD) runs a loop of 8
// independent chains of the instruction.  Build: hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(1024) void rate_kernel(long long* out, int iters, double seed) {
    __shared__ double tab[64 * 16];
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) tab[i] = 1.0 + i * 1e-3;
    __syncthreads();
    double a[8], b = seed, c = seed * 0.5;
    typedef double d4 __attribute__((__vector_size__(32)));
    d4 acc = {0, 0, 0, 0};
    int n[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3; n[i] = (int)threadIdx.x * 8 + i; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(n[i]));
#define MAXF(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define RND(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
#define ANDB(i) asm volatile("v_and_b32 %0, 63, %0" : "+v"(n[i]));
#define ASHR(i) asm volatile("v_ashrrev_i32 %0, 6, %0" : "+v"(n[i]));
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(n[i]));
#define SWAP(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(n[i]), "+v"(n[(i + 4) & 7]));
#define CMPF(i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(n[(i + 1) & 7]) : "vcc");
#define RCP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
#define FMAF(i) { float f = __int_as_float(n[i]); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f)); n[i] = __float_as_int(f); }
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
#define LDS(i) { int ad = (n[i] & 63) * 8; asm volatile("ds_read_b64 %0, %1" : "=v"(a[i]) : "v"(ad)); }
#define LDSW(i) asm volatile("s_waitcnt lgkmcnt(0)");
#define READL(i) { int s; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(n[i])); asm volatile("" :: "s"(s)); }
#define CVT(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(a[i]));
#define MAD64(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        if (OP == 0) { REP8(FMA) REP8(FMA) }
        if (OP == 1) { REP8(ADD) REP8(ADD) }
        if (OP == 2) { REP8(MUL) REP8(MUL) }
        if (OP == 3) { REP8(LDEXP) REP8(LDEXP) }
        if (OP == 4) { REP8(MAXF) REP8(MAXF) }
        if (OP == 5) { REP8(RND) REP8(RND) }
        if (OP == 6) { REP8(ADDU) REP8(ADDU) }
        if (OP == 7) { REP8(ANDB) REP8(ANDB) }
        if (OP == 8) { REP8(ASHR) REP8(ASHR) }
        if (OP == 9) { REP8(DPP) REP8(DPP) }
        if (OP == 10) { REP8(SWAP) REP8(SWAP) }
        if (OP == 11) { REP8(CMPF) REP8(CMPF) }
        if (OP == 12) { REP8(CNDM) REP8(CNDM) }
        if (OP == 13) { REP8(RCP) REP8(RCP) }
        if (OP == 14) { REP8(FMAF) REP8(FMAF) }
        if (OP == 15) { REP8(PKFMA) REP8(PKFMA) }
        if (OP == 16) { REP8(LDS) LDSW(0) REP8(LDS) LDSW(0) }
        if (OP == 17) { REP8(READL) REP8(READL) }
        if (OP == 18) { REP8(CVT) REP8(CVT) }
        if (OP == 19) { REP8(MAD64) REP8(MAD64) }
        if (OP == 20) { REP8(LSHLADD) REP8(LSHLADD) }
        if (OP == 21) { REP8(FMA) REP8(ADDU) }          // fp64 and int32 interleaved
        if (OP == 22) { REP8(FMA) REP8(LDS) LDSW(0) }   // fp64 and LDS reads interleaved
#define MFMA(i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a[i]), "v"(b));
        if (OP == 23) { REP8(MFMA) REP8(MFMA) }         // dependent accumulation chain
        if (OP == 24) { MFMA(0) REP8(FMA) MFMA(1) REP8(FMA) }   // one matrix instruction per 8 independent fp64 FMAs (count: 18 instead of 16)
    }
    const long long t1 = clock64();
    double s = 0; int m = 0;
    for (int i = 0; i < 8; ++i) { s += a[i]; m += n[i]; }
    if (s + acc[0] + acc[3] == 1234.5 && m == 77) out[1023] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

static const char* kNames[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_ldexp_f64", "v_max_f64", "v_rndne_f64", "v_add_u32",
                               "v_and_b32", "v_ashrrev_i32", "v_mov_b32_dpp", "v_permlane32_swap", "v_cmp_gt_f64", "v_cndmask_b32",
                               "v_rcp_f64", "v_fma_f32", "v_pk_fma_f32", "ds_read_b64 (8 in flight)", "v_readlane_b32",
                               "v_cvt_i32_f64", "v_mul_lo_u32", "v_lshl_add_u32", "fma_f64 + add_u32 (per pair)",
                               "fma_f64 + ds_read_b64 (per pair)", "v_mfma_f64_16x16x4 (dependent)",
                               "1 mfma_f64 + 8 fma_f64 (per 16 slots)"};

template <int OP>
void run(long long* dOut, int wavesPerSimd) {
    const int iters = 20000;
    const int threads = 64 * 4 * wavesPerSimd;          // a workgroup's waves are dealt round-robin to the CU's 4 SIMDs
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(1), dim3(threads), 0, 0, dOut, iters, 1.000001);      // warm
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(1), dim3(threads), 0, 0, dOut, iters, 1.000001);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(16);
    (void)hipMemcpy(h.data(), dOut, 16 * sizeof(long long), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
    const double perInstr = (double)mx / (iters * 16.0) / wavesPerSimd;
    std::printf("%-34s waves/SIMD %d: %7.2f clock64 ticks per wave instruction; kernel %8.1f us = %6.2f ns per instruction\n",
                kNames[OP], wavesPerSimd, perInstr, ms * 1e3, ms * 1e6 / (iters * 16.0) / wavesPerSimd);
}

template <int OP>
void run_all(long long* dOut) {
    run<OP>(dOut, 1);
    run<OP>(dOut, 2);
    run<OP>(dOut, 4);
    if constexpr (OP + 1 < 25) run_all<OP + 1>(dOut);
}

int main() {
    long long* dOut;
    (void)hipMalloc(&dOut, 1024 * sizeof(long long));
    (void)hipMemset(dOut, 0, 1024 * sizeof(long long));
    int clk = 0;
    (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0;
    (void)hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    std::printf("shader clock %d kHz, wall clock %d kHz\n", clk, wclk);
    // how many clock64 ticks per microsecond (clock64 may count a fixed-rate clock, not shader cycles)
    run_all<0>(dOut);
    return 0;
}
