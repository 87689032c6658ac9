// Probe: does a bricked arrangement of the cell-packed projector (4x4x4 cells = 4 KB contiguous) serve the E-step's
// gathers faster than the linear [z][y][x] arrangement?  Each wave mimics one (image, pixel sub-stream): its 64 lanes
// (= rotations of a particle-filter cloud) read cells within +-R voxels of a centre that walks 2 voxels per step along a
// random direction (= consecutive pixels of a slice).  64 bytes per lane per step, like interp_ft_packed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

constexpr int P = 512, NC = 257, NBX = 65, NB = 128;   // bricks: 65 x 128 x 128 of 4x4x4 cells

template <int LAYOUT>
__device__ __forceinline__ size_t cell_index(int x, int y, int z)
{
    if (LAYOUT == 0) return ((size_t)z * P + y) * NC + x;
    const size_t brick = ((size_t)(z >> 2) * NB + (y >> 2)) * NBX + (x >> 2);
    return brick * 64 + ((z & 3) << 4 | (y & 3) << 2 | (x & 3));
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ cells, float* __restrict__ out, int steps, int R, unsigned seed)
{
    const int lane = threadIdx.x & 63;
    const unsigned wid = (blockIdx.x * 4 + (threadIdx.x >> 6));
    const unsigned h0 = hash32(wid * 2654435761u + seed);
    // start point inside the sphere of radius 200 (x >= 0 half), direction in the unit cube
    float cx = (float)(h0 & 127) + 20.f, cy = (float)((h0 >> 7) & 255) + 128.f, cz = (float)((h0 >> 15) & 255) + 128.f;
    const unsigned h1 = hash32(h0 + 17u);
    float dx = (float)((int)(h1 & 255) - 128) / 128.f, dy = (float)((int)((h1 >> 8) & 255) - 128) / 128.f,
          dz = (float)((int)((h1 >> 16) & 255) - 128) / 128.f;
    const float dn = 2.0f / sqrtf(dx * dx + dy * dy + dz * dz + 1e-6f);
    dx *= dn; dy *= dn; dz *= dn;
    const unsigned hl = hash32(lane * 97u + h0);
    const int span = 2 * R + 1;
    const int ox = (int)(hl % span) - R, oy = (int)((hl >> 8) % span) - R, oz = (int)((hl >> 16) % span) - R;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        int x = (int)cx + ox, y = (int)cy + oy, z = (int)cz + oz;
        x = x < 0 ? 0 : (x > 255 ? 255 : x);
        y &= (P - 1); z &= (P - 1);
        const float4* c = cells + cell_index<LAYOUT>(x, y, z) * 4;
        const float4 a = c[0], b = c[1], d = c[2], e = c[3];
        acc += a.x + b.y + d.z + e.w;
        cx += dx; cy += dy; cz += dz;
        if (cx < 2.f || cx > 250.f) dx = -dx;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const size_t nCells = (size_t)NBX * NB * NB * 64;   // covers both layouts
    float4* cells; float* out;
    CK(hipMalloc(&cells, nCells * 64));
    CK(hipMemset(cells, 0, nCells * 64));
    const int nWG = 40000, steps = 200;
    CK(hipMalloc(&out, (size_t)nWG * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int Rs[] = {0, 1, 2, 4, 8};
    for (int ri = 0; ri < 5; ri++) {
        const int R = Rs[ri];
        for (int layout = 0; layout < 2; layout++) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                if (layout == 0) hipLaunchKernelGGL(k_gather<0>, dim3(nWG), dim3(256), 0, 0, cells, out, steps, R, 1234u + rep);
                else hipLaunchKernelGGL(k_gather<1>, dim3(nWG), dim3(256), 0, 0, cells, out, steps, R, 1234u + rep);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double bytes = (double)nWG * 256 * steps * 64.0;
            printf("R = %d  layout %s: %.2f ms  %.2f TB/s (algorithmic 64 B per lane-step)\n", R, layout ? "4x4x4 bricks" : "linear      ", best,
                   bytes / best / 1e9);
        }
    }
    return 0;
}
