// lds_atomic_bench.hip -- LDS accumulate rates on MI355X: ds_add_f32 vs ds_add_u32 vs plain read-add-write,
// random vs conflict-free addresses.  hipcc --offload-arch=gfx950 -O3 -o tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

constexpr int LDSN = 12288;  // floats (48 KB)

template <int MODE, int PATTERN>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters)
{
    __shared__ float s[LDSN];
    for (int i = threadIdx.x; i < LDSN; i += 256) s[i] = 0.f;
    __syncthreads();
    const uint32_t tid = threadIdx.x;
    uint32_t h = hash32(tid * 7919u + blockIdx.x);
    for (int it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        int idx;
        if (PATTERN == 0) idx = (int)((h >> 8) % LDSN);                       // random
        else if (PATTERN == 1) idx = (int)(((it * 256 + tid) * 1) % LDSN);   // conflict-free, distinct
        else if (PATTERN == 2) idx = (int)(((h >> 8) % 64) * 192 + (tid & 63));   // random rows, lane = bank (conflict-free)
        else idx = (int)(2 * ((it * 256 + tid) % (LDSN / 2)));                // distinct consecutive 8-byte slots (for MODE 4)
        const float v = (float)(it & 7) * 0.125f + 1.0f;
        if (MODE == 0) atomicAdd(&s[idx], v);                                        // ds_add_f32
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(s) + idx, (unsigned)(it + 1));  // ds_add_u32
        else if (MODE == 2) s[idx] += v;                                             // non-atomic RMW (racy; rate only)
        else if (MODE == 3) s[idx] = v;                                              // plain write
        else if (MODE == 4) {                                                        // 64-bit packed int add
            atomicAdd(reinterpret_cast<unsigned long long*>(s) + (idx >> 1), (unsigned long long)(it + 1));
        }
    }
    __syncthreads();
    float acc = 0.f;
    for (int i = threadIdx.x; i < LDSN; i += 256) acc += s[i];
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE, int PATTERN>
static void run(const char* name, float* out)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * 3, iters = 4096;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lds<MODE, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lds<MODE, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double n = (double)blocks * 256 * iters;
    printf("%-44s %8.2f ms  %9.1f G lane-ops/s  (%.2f lanes/clk/CU @2.4GHz)\n", name, ms, n / ms * 1e-6,
           n / ms * 1e-6 / 256 / 2.4);
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 4);
    run<0, 0>("ds_add_f32 random", out);
    run<0, 1>("ds_add_f32 linear (conflict-free)", out);
    run<0, 2>("ds_add_f32 random rows, lane=bank", out);
    run<1, 0>("ds_add_u32 random", out);
    run<1, 1>("ds_add_u32 linear", out);
    run<4, 0>("ds_add_u64 random", out);
    run<4, 3>("ds_add_u64 linear (conflict-free)", out);
    run<2, 0>("read-add-write random (non-atomic)", out);
    run<2, 1>("read-add-write linear (non-atomic)", out);
    run<3, 0>("ds_write_b32 random", out);
    run<3, 1>("ds_write_b32 linear", out);
    return 0;
}
