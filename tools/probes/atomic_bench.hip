// atomic_bench.hip -- micro-benchmark behind DESIGN.md section 5: fp32 global atomic-add rate on MI355X as a function
// of address pattern (random voxels vs runs of consecutive floats), to size the LDS-brick flush of the insert kernel.
// build: hipcc --offload-arch=gfx950 -O3 tools/atomic_bench.hip -o tools/atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// run = number of consecutive floats a group of `run` adjacent lanes hits; base of each run is random
template <int RUN>
__global__ void k_atomic(float* buf, size_t nElem, int iters)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = gid / RUN, off = gid % RUN;
    for (int it = 0; it < iters; it++) {
        const uint32_t h = hash32(grp * 2654435761U + it * 40503U);
        const size_t base = ((size_t)h % (nElem / RUN)) * RUN;
        unsafeAtomicAdd(&buf[base + off], 1.0f);
    }
}

template <int RUN>
__global__ void k_store(float* buf, size_t nElem, int iters)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = gid / RUN, off = gid % RUN;
    for (int it = 0; it < iters; it++) {
        const uint32_t h = hash32(grp * 2654435761U + it * 40503U);
        const size_t base = ((size_t)h % (nElem / RUN)) * RUN;
        buf[base + off] += 1.0f;  // non-atomic RMW for comparison
    }
}

template <typename K>
static void run(const char* name, K kern, float* buf, size_t nElem, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 16, threads = 256;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, buf, nElem, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, buf, nElem, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    double n = (double)blocks * threads * iters;
    printf("%-28s %8.2f ms  %8.2f G ops/s\n", name, ms, n / ms * 1e-6);
}

int main()
{
    const size_t nElem = (size_t)192 * 1024 * 1024;  // 768 MiB of floats, > MALL
    float* buf;
    hipMalloc(&buf, nElem * sizeof(float));
    hipMemset(buf, 0, nElem * sizeof(float));
    const int iters = 64;
    run("atomic run=1 (random)", k_atomic<1>, buf, nElem, iters);
    run("atomic run=2", k_atomic<2>, buf, nElem, iters);
    run("atomic run=4", k_atomic<4>, buf, nElem, iters);
    run("atomic run=8", k_atomic<8>, buf, nElem, iters);
    run("atomic run=16", k_atomic<16>, buf, nElem, iters);
    run("atomic run=32", k_atomic<32>, buf, nElem, iters);
    run("atomic run=64", k_atomic<64>, buf, nElem, iters);
    run("rmw    run=1 (random)", k_store<1>, buf, nElem, iters);
    run("rmw    run=16", k_store<16>, buf, nElem, iters);
    run("rmw    run=64", k_store<64>, buf, nElem, iters);
    // small footprint (fits L2/MALL): 8 MiB
    run("atomic run=1, 8MiB", k_atomic<1>, buf, (size_t)2 * 1024 * 1024, iters);
    run("atomic run=16, 8MiB", k_atomic<16>, buf, (size_t)2 * 1024 * 1024, iters);
    hipFree(buf);
    return 0;
}
