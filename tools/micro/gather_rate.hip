// micro-benchmark: L1 (TCP) throughput of scattered 64-byte record gathers, the access pattern of a BVH
// node fetch.  A: every lane loads its own record with 4 x dwordx4.  B: the same 64 records per wave are
// loaded cooperatively (4 adjacent lanes read the 4 16-byte chunks of one record -> one 64-byte request)
// and handed to their owners through LDS.  C: like B without the LDS hand-over (upper bound).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const float4 *rec, uint32_t nrec_mask, int iters, float *out, int active_mod)
{
    __shared__ float4 sh[256 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wbase = tid & ~63;
    uint32_t s = mix(blockIdx.x * 256 + tid + 1);
    float acc = 0.0f;
    const bool active = (lane % 8) < active_mod;           // emulate partial lane utilisation
    for (int it = 0; it < iters; it++) {
        s = mix(s + it);
        const uint32_t idx = active ? (s & nrec_mask) : 0u;
        if (MODE == 0) {
            if (active) {
                const float4 *p = rec + (size_t)idx * 4;
                float4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x + b.y + c.z + d.w;
            }
        } else {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t other = __shfl(idx, (lane >> 2) + 16 * k, 64);
                v[k] = rec[(size_t)other * 4 + (lane & 3)];
            }
            if (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 4; k++) sh[(wbase + (lane >> 2) + 16 * k) * 4 + (lane & 3)] = v[k];
                // same wave: LDS ops are in order, no barrier needed
                const float4 a = sh[tid * 4 + 0], b = sh[tid * 4 + 1], c = sh[tid * 4 + 2], d = sh[tid * 4 + 3];
                if (active) acc += a.x + b.y + c.z + d.w;
            } else {
                acc += v[0].x + v[1].y + v[2].z + v[3].w;
            }
        }
    }
    out[blockIdx.x * 256 + tid] = acc;
}
int main(int argc, char **argv)
{
    const uint32_t nrec = 1u << (argc > 1 ? atoi(argv[1]) : 17);      // 2^17 records x 64 B = 8 MB (L2 / MALL resident like the BVH)
    printf("records: %u (%.1f MB)\n", nrec, nrec * 64.0 / 1048576.0);
    float4 *rec; float *out;
    (void)hipMalloc(&rec, (size_t)nrec * 64); (void)hipMemset(rec, 0, (size_t)nrec * 64);
    const int blocks = 1536, iters = 2000;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int am : {8, 4}) for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(256), 0, 0, rec, nrec - 1, iters, out, am);
            if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, rec, nrec - 1, iters, out, am);
            if (mode == 2) hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(256), 0, 0, rec, nrec - 1, iters, out, am);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double recs = (double)blocks * 256 * iters * am / 8.0;
        printf("active %d/8 mode %d: %.3f ms  %.2f Grecords/s  (%.2f clk per record per CU at 2.4 GHz)\n", am, mode, best, recs / best / 1e6,
               best * 1e-3 * 2.4e9 * 256 / recs);
    }
    return 0;
}
