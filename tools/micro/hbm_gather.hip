// micro-benchmark (round 6): what rocprofv3's FETCH_SIZE reports for RANDOM RECORD GATHERS -- k_trace's access pattern -- against a known byte
// count, at working sets from L2-resident to far beyond the 256 MiB Infinity Cache.  The guide calibrates "FETCH_SIZE x 2" on a coalesced stream only and says
// "calibrate on a known byte count in your own access pattern"; bench.py's big-scene roofline used the stream factor for gathers.
//   stream   : every lane reads 16 bytes, consecutive lanes consecutive addresses, the whole working set once          known: WS bytes
//   gather64 : every lane reads ONE random 64-byte record (4 x global_load_dwordx4, as a node fetch does), R per lane      known: 64 B per record (the 128-byte line it lies in: 128 B)
//   gather128: every lane reads ONE random 128-byte record (8 x dwordx4)                                                  known: 128 B per record
// One kernel instantiation per (kind, working-set id), so that a rocprofv3 --kernel-trace --pmc pass reports each by name.  Run it plain for the rates, then under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./hbm_gather ;  ... --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum -- ./hbm_gather
// tools/hbm_gather_summary.py joins the three outputs.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_gather hbm_gather.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int KIND, int WSID>
__global__ __launch_bounds__(256) void k_hbm(const float4 *mem, uint64_t n16, uint32_t rec_mask, int per_lane, float *out)
{
    const uint64_t gtid = (uint64_t)blockIdx.x * 256 + threadIdx.x, gsz = (uint64_t)gridDim.x * 256;
    float acc = 0.0f;
    if (KIND == 0) {
        for (uint64_t i = gtid; i < n16; i += gsz) { const float4 v = mem[i]; acc += v.x + v.w; }
    } else {
        uint32_t s = mix((uint32_t)gtid * 2654435761u + 12345u);
        for (int it = 0; it < per_lane; it++) {
            s = mix(s + (uint32_t)it * 0x9e3779b9u);
            const uint32_t idx = s & rec_mask;
            const float4 *p = mem + (size_t)idx * (KIND == 1 ? 4 : 8);
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
            if (KIND == 2) { const float4 e = p[4], f = p[5], g = p[6], h = p[7]; acc += e.x + f.y + g.z + h.w; }
        }
    }
    out[gtid] = acc;
}
template <int KIND, int WSID> static void run(const char *kind, const float4 *mem, uint64_t ws_bytes, float *out, int cus)
{
    const int blocks = cus * 8, per_lane = 64;
    const uint64_t nrec = ws_bytes / (KIND == 2 ? 128 : 64);           // a power of two
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_hbm<KIND, WSID>), dim3(blocks), dim3(256), 0, 0, mem, ws_bytes / 16, (uint32_t)(nrec - 1), per_lane, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double recs = (double)blocks * 256 * per_lane;
    const double bytes = KIND == 0 ? (double)ws_bytes : recs * (KIND == 1 ? 64.0 : 128.0);
    printf("%-9s ws %8.0f MB  kernel k_hbm<%d,%d>  launches 3  %s %.0f  known_bytes %.0f  best %.4f ms  %.1f GB/s of known bytes%s\n", kind, ws_bytes / 1048576.0, KIND, WSID,
           KIND == 0 ? "lanes16B" : "records", KIND == 0 ? (double)ws_bytes / 16 : recs, bytes, best, bytes / best / 1e6,
           KIND == 1 ? " (x 2 in 128-byte lines)" : "");
}
int main()
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    const uint64_t max_ws = (uint64_t)8 << 30;
    float4 *mem; float *out;
    if (hipMalloc(&mem, max_ws) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(mem, 0, max_ws);
    (void)hipMalloc(&out, sizeof(float) * (size_t)cus * 8 * 256);
    printf("%s, %d CUs; working sets 2 MB .. 8 GB (L2 4 MiB per XCD, Infinity Cache 256 MiB)\n", pr.gcnArchName, cus);
#define ALL(WSID, BYTES) run<0, WSID>("stream", mem, (uint64_t)(BYTES), out, cus); run<1, WSID>("gather64", mem, (uint64_t)(BYTES), out, cus); run<2, WSID>("gather128", mem, (uint64_t)(BYTES), out, cus);
    ALL(0, 2ull << 20) ALL(1, 32ull << 20) ALL(2, 128ull << 20) ALL(3, 512ull << 20) ALL(4, 2ull << 30) ALL(5, 8ull << 30)
    return 0;
}
