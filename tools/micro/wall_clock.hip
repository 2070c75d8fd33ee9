// wall_clock64() rate on this device: the attribute, and ticks counted by a spinning kernel against HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long *out, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    out[0] = wall_clock64() - t0;
}
int main()
{
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    unsigned long long *d; hipMalloc(&d, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, d, 1000ull);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, d, 1000000ull);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("hipDeviceAttributeWallClockRate %d kHz; %llu ticks in %.3f ms = %.2f MHz\n", khz, h, ms, (double)h / ms / 1e3);
    return 0;
}
