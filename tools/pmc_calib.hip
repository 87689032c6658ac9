// pmc_calib.hip -- dispatches of KNOWN byte count for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Three kernels, 1 GiB each, from arrays far larger than L2 + Infinity Cache, every byte touched once:
//   k_cal_stream      1 GiB read + 1 GiB written, float4 per lane, coalesced (the guide's own calibration pattern)
//   k_cal_gather64    2^24 lanes, each reads ONE distinct, pseudo-randomly placed 64-byte cell (4 x float4 at consecutive
//                     addresses, 64-byte aligned) of an 8 GiB array -- the access shape of interp_ft_packed (E-step)
//   k_cal_gather16    2^26 lanes, each reads ONE distinct, pseudo-randomly placed 16-byte pair of an 8 GiB array -- the
//                     row-pair reads of interp_ft on the standard volume layout
// hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib tools/pmc_calib.hip ; run under rocprofv3 --pmc (tools/pmc_traffic.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_cal_stream(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// cell index = (i * odd) mod 2^27: a bijection on the 2^27 cells of the array, consecutive lanes 2.6e9 cells apart
__global__ __launch_bounds__(256) void k_cal_gather64(const float4* __restrict__ cells, float* __restrict__ out, unsigned nLane)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nLane) return;
    const unsigned c = (i * 0x9E3779B1u) & ((1u << 27) - 1u);
    const float4* p = cells + (size_t)c * 4;
    const float4 a = p[0], b = p[1], d = p[2], e = p[3];
    const float s = a.x + b.y + d.z + e.w;
    if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void k_cal_gather16(const float4* __restrict__ pairs, float* __restrict__ out, unsigned nLane)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nLane) return;
    const unsigned c = (i * 0x9E3779B1u) & ((1u << 29) - 1u);   // 2^29 pairs of 16 B = 8 GiB
    const float4 a = pairs[c];
    if (a.x + a.w == 12345.678f) out[0] = a.x;
}

int main()
{
    const size_t GiB = 1ull << 30;
    float4 *big, *src, *dst;
    float* out;
    CK(hipMalloc(&big, 8 * GiB));
    CK(hipMalloc(&src, GiB));
    CK(hipMalloc(&dst, GiB));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(big, 0, 8 * GiB));
    CK(hipMemset(src, 0, GiB));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        const size_t n4 = GiB / 16;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_cal_stream, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, src, dst, n4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_cal_stream   1 GiB in + 1 GiB out : %.3f ms  %.2f TB/s\n", ms, 2.0 * GiB / ms / 1e9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_cal_gather64, dim3((1u << 24) / 256), dim3(256), 0, 0, big, out, 1u << 24);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_cal_gather64 1 GiB of 64-B cells   : %.3f ms  %.2f TB/s  %.1f G requests/s\n", ms, 1.0 * GiB / ms / 1e9, (1u << 24) / ms / 1e6);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_cal_gather16, dim3((1u << 26) / 256), dim3(256), 0, 0, big, out, 1u << 26);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_cal_gather16 1 GiB of 16-B pairs   : %.3f ms  %.2f TB/s  %.1f G requests/s\n", ms, 1.0 * GiB / ms / 1e9, (1u << 26) / ms / 1e6);
    }
    return 0;
}
