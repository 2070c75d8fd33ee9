// micro-benchmark: rate of wave-aggregated returning atomics on ONE address (the dense-queue
// append pattern of k_shade / the ray re-fetch of k_trace).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_same(int *ctr, int *out, int n, int mode)
{
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    int lane = threadIdx.x & 63;
    int base = 0;
    if (mode == 0) base = tid & ~63;                                      // no atomic
    else if (mode == 1) { if (lane == 0) base = atomicAdd(ctr, 64); base = __shfl(base, 0, 64); }              // one address
    else if (mode == 2) { if (lane == 0) base = atomicAdd(ctr + 64 * (blockIdx.x & 7), 64); base = __shfl(base, 0, 64) + (blockIdx.x & 7) * (n / 8); }  // 8 addresses
    else if (mode == 3) {                                                 // block-aggregated through LDS
        __shared__ int sb;
        if (threadIdx.x == 0) sb = atomicAdd(ctr, (int)blockDim.x);
        __syncthreads();
        base = sb + (threadIdx.x & ~63);
    }
    else if (mode == 4) { if (lane == 0) atomicAdd(ctr, 64); base = tid & ~63; }   // non-returning
    if (base + lane < n) out[base + lane] = tid;
}
int main()
{
    const int n = 32 << 20;
    int *ctr, *out; hipMalloc(&ctr, 4096); hipMalloc(&out, sizeof(int) * (size_t)n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bs : {256, 1024}) for (int mode = 0; mode < 5; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipMemset(ctr, 0, 4096);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_same, dim3(n / bs), dim3(bs), 0, 0, ctr, out, n, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("block %4d mode %d: %.3f ms  (%d waves -> %.2f ns/wave-atomic)\n", bs, mode, best, n / 64, best * 1e6 / (n / 64));
    }
    return 0;
}
