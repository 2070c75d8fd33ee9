// micro-benchmark (gfx950): what a scattered record gather costs the L1 / texture-address path of a CU, by HOW the record is read --
// the question behind k_trace's node format (profiles/r05_bound_ladder.txt: the kernel slows down by what a visit's loads cost, not by its VALU work).
// Every lane reads its own random 64-byte-aligned record; modes: N x global_load_dwordx4 (16 N bytes), N x dwordx2, N x dword of the same line,
// so that cost per REQUEST and cost per BYTE separate.  Reported: clocks per record per CU (event time x clock from s_memtime / wall clock), Grecords/s.
//   hipcc --offload-arch=gfx950 -O3 -o ta_cost ta_cost.hip && ./ta_cost [log2 records]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int WIDTH, int N>          // WIDTH dwords per load (1, 2, 4), N loads per record
__global__ __launch_bounds__(256) void k(const uint32_t *rec, uint32_t mask, int iters, float *out)
{
    uint32_t s = mix(blockIdx.x * 256 + threadIdx.x + 1);
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        s = mix(s + it);
        const uint32_t *p = rec + (size_t)(s & mask) * 16;
#pragma unroll
        for (int j = 0; j < N; j++) {
            if (WIDTH == 4) { const uint4 v = *(const uint4 *)(p + 4 * j); acc += v.x ^ v.y ^ v.z ^ v.w; }
            if (WIDTH == 2) { const uint2 v = *(const uint2 *)(p + 4 * j); acc += v.x ^ v.y; }
            if (WIDTH == 1) { acc += p[4 * j]; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
}
// 128-byte records (two adjacent 64-byte lines, 128-byte aligned): does the second half cost a second gather?
template <int N>          // N x dwordx4 from the start of the record
__global__ __launch_bounds__(256) void k128(const uint32_t *rec, uint32_t mask, int iters, float *out)
{
    uint32_t s = mix(blockIdx.x * 256 + threadIdx.x + 1);
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        s = mix(s + it);
        const uint32_t *p = rec + (size_t)(s & mask) * 32;
#pragma unroll
        for (int j = 0; j < N; j++) { const uint4 v = *(const uint4 *)(p + 4 * j); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
}
template <int N> void run128(const uint32_t *rec, uint32_t nrec128, float *out, int cus)
{
    const int blocks = cus * 6, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k128<N>), dim3(blocks), dim3(256), 0, 0, rec, nrec128 - 1, iters, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double recs = (double)blocks * 256 * iters;
    printf("  128-byte records, %d x dwordx4 (%3d B read): %7.3f ms  %6.1f Grecords/s  %5.2f ns per record per CU\n", N, 16 * N, best, recs / best / 1e6, best * 1e6 * cus / recs);
}
template <int WIDTH, int N> void run(const uint32_t *rec, uint32_t nrec, float *out, int cus)
{
    const int blocks = cus * 6, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<WIDTH, N>), dim3(blocks), dim3(256), 0, 0, rec, nrec - 1, iters, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double recs = (double)blocks * 256 * iters;
    printf("  %d x %-8s (%2d B of the record): %7.3f ms  %6.1f Grecords/s  %5.2f ns per record per CU  %6.1f Grequests/s\n", N,
           WIDTH == 4 ? "dwordx4" : (WIDTH == 2 ? "dwordx2" : "dword"), 4 * WIDTH * N, best, recs / best / 1e6, best * 1e6 * cus / recs, recs * N / best / 1e6);
}
int main(int argc, char **argv)
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float *out; (void)hipMalloc(&out, sizeof(float) * (size_t)cus * 6 * 256);
    for (int lg : {15, 17}) {
        if (argc > 1) lg = atoi(argv[1]);
        const uint32_t nrec = 1u << lg;
        uint32_t *rec; (void)hipMalloc(&rec, (size_t)nrec * 64); (void)hipMemset(rec, 1, (size_t)nrec * 64);
        printf("records: %u x 64 B = %.1f MB, 6 blocks of 256 threads per CU\n", nrec, nrec * 64.0 / 1048576.0);
        run<4, 4>(rec, nrec, out, cus); run<4, 3>(rec, nrec, out, cus); run<4, 2>(rec, nrec, out, cus); run<4, 1>(rec, nrec, out, cus);
        run<2, 4>(rec, nrec, out, cus); run<2, 2>(rec, nrec, out, cus); run<2, 1>(rec, nrec, out, cus);
        run<1, 4>(rec, nrec, out, cus); run<1, 2>(rec, nrec, out, cus); run<1, 1>(rec, nrec, out, cus);
        run128<8>(rec, nrec / 2, out, cus); run128<6>(rec, nrec / 2, out, cus); run128<5>(rec, nrec / 2, out, cus); run128<4>(rec, nrec / 2, out, cus);
        (void)hipFree(rec);
        if (argc > 1) break;
    }
    return 0;
}
