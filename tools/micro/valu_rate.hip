// micro-benchmark: issue rate of the VALU instructions k_trace's node step is made of (gfx950).  Each kernel runs a long chain of
// 8 independent accumulators of ONE instruction kind; reported: cycles per wave-instruction per SIMD at full occupancy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b, unsigned u)
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    unsigned h0 = threadIdx.x * 2654435761u, h1 = h0 ^ 0x12345u, h2 = h0 + 77u, h3 = h0 * 3u;
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));) }
        if (KIND == 1) { REP16(asm volatile("v_fma_mix_f32 %0, %10, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %11, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %12, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %13, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %10, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %11, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %12, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %13, %8, %9 op_sel_hi:[1,0,0]" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "v"(h0), "v"(h1), "v"(h2), "v"(h3));) }
        if (KIND == 2) { REP16(asm volatile("v_cvt_f32_u32_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %1, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %2, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %3, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %5, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %6, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %7, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(h0), "v"(h1), "v"(h2), "v"(h3));) }
        if (KIND == 3) { REP16(asm volatile("v_alignbit_b32 %0, %0, %0, %4\n v_alignbit_b32 %1, %1, %1, %4\n v_alignbit_b32 %2, %2, %2, %4\n v_alignbit_b32 %3, %3, %3, %4\n v_alignbit_b32 %0, %0, %0, %4\n v_alignbit_b32 %1, %1, %1, %4\n v_alignbit_b32 %2, %2, %2, %4\n v_alignbit_b32 %3, %3, %3, %4" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(u));) }
        if (KIND == 4) { REP16(asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));) }
        if (KIND == 5) { REP16(asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));) }
        if (KIND == 6) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");) }
        if (KIND == 7) { REP16(asm volatile("v_cmp_lt_f32 s[10:11], %0, %8\n v_cndmask_b32 %0, %0, %1, s[10:11]\n v_cmp_lt_f32 s[12:13], %2, %8\n v_cndmask_b32 %2, %2, %3, s[12:13]\n v_cmp_lt_f32 s[14:15], %4, %8\n v_cndmask_b32 %4, %4, %5, s[14:15]\n v_cmp_lt_f32 s[16:17], %6, %8\n v_cndmask_b32 %6, %6, %7, s[16:17]" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (float)(h0 ^ h1 ^ h2 ^ h3);
}
template <int KIND> void run(const char *name, float *out)
{
    const int blocks = 256 * 8, iters = 2000;          // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, 16u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double insts_per_simd = (double)blocks * 4 /*waves*/ * iters * 128.0 / 1024.0;   // wave-instructions per SIMD
    printf("%-28s %.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, best, best * 1e-3 * 2.4e9 / insts_per_simd);
}
int main()
{
    float *out; (void)hipMalloc(&out, sizeof(float) * 256 * 8 * 256);
    run<0>("v_fma_f32", out); run<1>("v_fma_mix_f32 (f16 src)", out); run<2>("v_cvt_f32_u32_sdwa", out); run<3>("v_alignbit_b32", out);
    run<4>("v_max3_f32", out); run<5>("v_min_f32", out); run<6>("v_cmp(vcc)+v_cndmask pair /2", out); run<7>("v_cmp(sgpr)+v_cndmask pair /2", out);
    return 0;
}
