// Probe: the rate of scattered 64-byte gathers (the E-step's cell-packed requests) as a function of the FOOTPRINT they fall in:
// within the 4 MiB L2 of an XCD, within the 256 MiB Infinity Cache, beyond it.  If footprints that fit the Infinity Cache are served
// much faster than the 50 G requests/s the E-step sees, a schedule that keeps view-neighbouring images on the same region of the
// volume at the same time has something to gain; if not, the wall is the fabric's request rate whatever the DRAM does.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mall_probe tools/mall_probe.hip && tools/mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// every lane reads `steps` random 64-byte cells of the first nCells cells (nCells a power of two)
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ cells, float* __restrict__ out, int steps, unsigned mask, unsigned seed)
{
    unsigned h = hash32((blockIdx.x * 256u + threadIdx.x) * 2654435761u + seed);
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        h = hash32(h + 0x9e3779b9u);
        const float4* c = cells + (size_t)(h & mask) * 4;
        const float4 a = c[0], b = c[1], d = c[2], e = c[3];
        acc += a.x + b.y + d.z + e.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const size_t maxBytes = (size_t)8 << 30;
    float4* cells; float* out;
    CK(hipMalloc(&cells, maxBytes));
    CK(hipMemset(cells, 0, maxBytes));
    const int nWG = 16384, steps = 256;
    CK(hipMalloc(&out, (size_t)nWG * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int lg = 21; lg <= 33; lg++) {   // 2 MiB ... 8 GiB
        const size_t bytes = (size_t)1 << lg;
        const unsigned mask = (unsigned)(bytes / 64 - 1);
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather, dim3(nWG), dim3(256), 0, 0, cells, out, steps, mask, 77u + rep);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double req = (double)nWG * 256 * steps;
        printf("footprint %8.0f MiB: %7.2f ms  %6.1f G requests/s  %5.2f TB/s\n", bytes / 1048576.0, best, req / best / 1e6, req * 64 / best / 1e9);
    }
    return 0;
}
