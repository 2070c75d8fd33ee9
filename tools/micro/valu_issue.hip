// micro-benchmark (gfx950): how many SHADER CYCLES a SIMD needs per wave64 VALU instruction, by instruction kind and by waves per SIMD.
// Unlike valu_rate.hip (round 2: wall time x an assumed 2.4 GHz) the cycles are read on the device (s_memtime at the start and the end of every
// wave; span = last end - first start), so the result does not depend on the clock the chip happens to sustain; the clock is reported too
// (span cycles / event time).  Each kernel is a chain of 8 independent accumulators of ONE instruction kind, 128 instructions per iteration.
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip && ./valu_issue
// Under rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE the same launches calibrate bench.py's "valu busy".
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define A8(ins, tail) ins " %0, %0" tail "\n" ins " %1, %1" tail "\n" ins " %2, %2" tail "\n" ins " %3, %3" tail "\n" ins " %4, %4" tail "\n" ins " %5, %5" tail "\n" ins " %6, %6" tail "\n" ins " %7, %7" tail
#define OUT8 "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
typedef float f2v __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *ticks, int iters, float a, float b, unsigned u, unsigned long long m)
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    f2v p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, p4 = {r1, r0}, p5 = {r3, r2}, p6 = {r5, r4}, p7 = {r7, r6}, pa = {a, a}, pb = {b, b};
    const float sa = __builtin_amdgcn_readfirstlane(a);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0)  { REP16(asm volatile(A8("v_fma_f32", ", %8, %9") : OUT8 : "v"(a), "v"(b));) }            // three VGPR sources
        if (KIND == 1)  { REP16(asm volatile(A8("v_fma_f32", ", 1.0, %8") : OUT8 : "s"(sa));) }                  // one VGPR source
        if (KIND == 2)  { REP16(asm volatile(A8("v_fma_f32", ", %8, 0.5") : OUT8 : "v"(a));) }                   // two VGPR sources
        if (KIND == 3)  { REP16(asm volatile(A8("v_add_f32", ", %8") : OUT8 : "v"(a));) }
        if (KIND == 4)  { REP16(asm volatile(A8("v_mul_f32", ", %8") : OUT8 : "v"(a));) }
        if (KIND == 5)  { REP16(asm volatile(A8("v_pk_fma_f32", ", %8, %9") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));) }
        if (KIND == 6)  { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : OUT8);) }
        if (KIND == 7)  { REP16(asm volatile(A8("v_fma_mix_f32", ", %8, %9 op_sel_hi:[1,0,0]") : OUT8 : "v"(a), "v"(b));) }
        if (KIND == 8)  { REP16(asm volatile(A8("v_alignbit_b32", ", %0, %8") : OUT8 : "v"(u));) }
        if (KIND == 9)  { REP16(asm volatile(A8("v_min_f32", ", %8") : OUT8 : "v"(a));) }
        if (KIND == 10) { REP16(asm volatile(A8("v_cndmask_b32", ", %8, %9") : OUT8 : "v"(a), "s"(m));) }
        if (KIND == 11) { REP16(asm volatile(A8("v_max3_f32", ", %8, %9") : OUT8 : "v"(a), "v"(b));) }
        if (KIND == 12) { REP16(asm volatile(A8("v_and_b32", ", %8") : OUT8 : "v"(u));) }
        if (KIND == 13) { REP16(asm volatile(A8("v_add_u32", ", %8") : OUT8 : "v"(u));) }
        if (KIND == 14) { REP16(asm volatile(A8("v_min3_i32", ", %8, %9") : OUT8 : "v"(a), "v"(b));) }
        // round 6: the packed-fp16 instructions a two-children-at-once box test would be made of, and what would surround them
        if (KIND == 16) { REP16(asm volatile(A8("v_pk_fma_f16", ", %8, %9") : OUT8 : "v"(a), "v"(b));) }
        if (KIND == 17) { REP16(asm volatile(A8("v_pk_min_f16", ", %8") : OUT8 : "v"(a));) }
        if (KIND == 18) { REP16(asm volatile(A8("v_pk_max_f16", ", %8") : OUT8 : "v"(a));) }
        if (KIND == 19) { REP16(asm volatile(A8("v_pk_add_f16", ", %8 neg_lo:[0,1] neg_hi:[0,1]") : OUT8 : "v"(a));) }
        if (KIND == 20) { REP16(asm volatile(A8("v_perm_b32", ", %8, %9") : OUT8 : "v"(a), "v"(u));) }
        if (KIND == 21) { REP16(asm volatile(A8("v_cvt_f32_f16", "") : OUT8);) }
        if (KIND == 22) { REP16(asm volatile(A8("v_min_u32", ", %8") : OUT8 : "v"(u));) }
        if (KIND == 23) { REP16(asm volatile(A8("v_pk_min_u16", ", %8") : OUT8 : "v"(u));) }
        if (KIND == 24) { REP16(asm volatile(A8("v_pk_ashrrev_i16", ", 15") : OUT8);) }
        if (KIND == 25) { REP16(asm volatile(A8("v_or_b32", ", %8") : OUT8 : "v"(u));) }
        if (KIND == 26) { REP16(asm volatile(A8("v_lshl_or_b32", ", %8, %9") : OUT8 : "v"(u), "v"(a));) }
        if (KIND == 27) { REP16(asm volatile(A8("v_bfi_b32", ", %8, %9") : OUT8 : "v"(u), "v"(a));) }
        if (KIND == 28) { REP16(asm volatile(A8("v_med3_f32", ", %8, %9") : OUT8 : "v"(a), "v"(b));) }
        if (KIND == 29) { REP16(asm volatile(A8("v_max_f32", ", %8") : OUT8 : "v"(a));) }
        if (KIND == 30) { REP16(asm volatile(A8("v_min_i32", ", %8") : OUT8 : "v"(u));) }
        if (KIND == 31) { REP16(asm volatile(A8("v_pk_mul_f16", ", %8") : OUT8 : "v"(a));) }
        if (KIND == 15) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : OUT8 : "v"(a) : "vcc");) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) { const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); ticks[2 * w] = t0; ticks[2 * w + 1] = t1; }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int KIND> void run(const char *name, float *out, unsigned long long *ticks, int cus)
{
    const int iters = 1000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-34s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {               // waves per SIMD: `wps` 256-thread blocks per CU
        const int blocks = cus * wps;
        double best_ms = 0, own_cpi = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters, 1.0001f, 0.5f, 16u, 0x5555555555555555ull);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> t(2 * (size_t)blocks * 4);
            (void)hipMemcpy(t.data(), ticks, t.size() * 8, hipMemcpyDeviceToHost);
            double own = 0;
            for (size_t w = 0; w < (size_t)blocks * 4; w++) own += (double)(t[2 * w + 1] - t[2 * w]);
            if (ms < best_ms || best_ms == 0) { best_ms = ms; own_cpi = own / ((double)blocks * 4) / (iters * 128.0); }
        }
        // a wave's own s_memtime ticks per instruction / waves per SIMD = ticks per instruction of the SIMD; the event time gives ns per instruction of the SIMD
        const double ns = best_ms * 1e6 / ((double)wps * iters * 128.0);
        printf("  w%d: %5.2f tick (own %5.2f) %5.3f ns", wps, own_cpi / wps, own_cpi, ns);
    }
    printf("\n");
}
int main()
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float *out; (void)hipMalloc(&out, sizeof(float) * (size_t)cus * 8 * 256);
    unsigned long long *ticks; (void)hipMalloc(&ticks, 16 * (size_t)cus * 8 * 4);
    printf("%s, %d CUs; per wave64 instruction and SIMD: s_memtime ticks (own ticks of a wave per instruction / waves per SIMD) and ns (event time), by waves per SIMD\n", pr.gcnArchName, cus);
    run<0>("v_fma_f32 v,v,v,v", out, ticks, cus); run<1>("v_fma_f32 v,v,1.0,s", out, ticks, cus); run<2>("v_fma_f32 v,v,v,0.5", out, ticks, cus);
    run<3>("v_add_f32", out, ticks, cus); run<4>("v_mul_f32", out, ticks, cus); run<5>("v_pk_fma_f32", out, ticks, cus); run<6>("v_mov_b32", out, ticks, cus);
    run<7>("v_fma_mix_f32 (f16 src0)", out, ticks, cus); run<8>("v_alignbit_b32", out, ticks, cus); run<9>("v_min_f32", out, ticks, cus);
    run<10>("v_cndmask_b32 (sgpr mask)", out, ticks, cus); run<11>("v_max3_f32", out, ticks, cus); run<12>("v_and_b32", out, ticks, cus); run<13>("v_add_u32", out, ticks, cus);
    run<14>("v_min3_i32", out, ticks, cus); run<15>("v_cmp_lt_f32 vcc", out, ticks, cus);
    run<16>("v_pk_fma_f16", out, ticks, cus); run<17>("v_pk_min_f16", out, ticks, cus); run<18>("v_pk_max_f16", out, ticks, cus); run<19>("v_pk_add_f16 (neg src1)", out, ticks, cus);
    run<31>("v_pk_mul_f16", out, ticks, cus); run<20>("v_perm_b32", out, ticks, cus); run<21>("v_cvt_f32_f16", out, ticks, cus); run<22>("v_min_u32", out, ticks, cus); run<30>("v_min_i32", out, ticks, cus);
    run<23>("v_pk_min_u16", out, ticks, cus); run<24>("v_pk_ashrrev_i16", out, ticks, cus); run<25>("v_or_b32", out, ticks, cus); run<26>("v_lshl_or_b32", out, ticks, cus);
    run<27>("v_bfi_b32", out, ticks, cus); run<28>("v_med3_f32", out, ticks, cus); run<29>("v_max_f32", out, ticks, cus);
    return 0;
}
