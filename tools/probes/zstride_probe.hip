// zstride_probe.hip -- why does the gridding loop's z pass run at 2.2 TB/s at 1 024^3 when the y pass reaches 4.9?
// Reads and rewrites C [P][P][ncp] complex64 along z exactly as k_fft_z_update does (tiles of TX = 8 adjacent x columns = 64-byte
// segments, one per z plane), with no arithmetic, in two layouts of the z index:
//   natural  element kz at kz * P * ncp                      (a 4.2 MB stride at P = 1024: every segment on another 2 MB page)
//   blocked  element kz at (kz / B) * (P * B * ncp) + (kz % B) * ncp + ky * B * ncp   (B planes interleaved row by row: B segments per page run)
// and the y pass's pattern for reference.  build: hipcc --offload-arch=gfx950 -O3 -o zstride_probe zstride_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(1024) void k_sweep(float2* C, int P, int ncp, int B)
{
    const int TX = 8, c = threadIdx.x % TX, t = threadIdx.x / TX;   // t < 128
    const int x = blockIdx.x * TX + c, jw = blockIdx.y;
    if (x >= P / 2 + 1) return;
    float2 v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const long e = t + 128 * q;
        long off;
        if (MODE == 0) off = e * (long)P * ncp + (long)jw * ncp + x;                                   // z, natural
        else if (MODE == 1) off = (e / B) * ((long)P * B * ncp) + ((long)jw * B + (e % B)) * ncp + x;   // z, blocked
        else off = (long)jw * P * ncp + e * ncp + x;                                                    // y
        v[q] = C[off];
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const long e = t + 128 * q;
        long off;
        if (MODE == 0) off = e * (long)P * ncp + (long)jw * ncp + x;
        else if (MODE == 1) off = (e / B) * ((long)P * B * ncp) + ((long)jw * B + (e % B)) * ncp + x;
        else off = (long)jw * P * ncp + e * ncp + x;
        C[off] = make_float2(v[q].x + 1.f, v[q].y);
    }
}
int main(int argc, char** argv)
{
    const int P = 1024, ncp = 520;
    const size_t n = (size_t)P * P * ncp;
    float2* C;
    if (hipMalloc(&C, n * sizeof(float2)) != hipSuccess) return 1;
    hipMemset(C, 0, n * sizeof(float2));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const dim3 g((P / 2 + 1 + 7) / 8, P), blk(1024);
    auto run = [&](int mode, int B, const char* name) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            for (int i = 0; i < 5; i++) {
                if (mode == 0) hipLaunchKernelGGL(k_sweep<0>, g, blk, 0, 0, C, P, ncp, B);
                else if (mode == 1) hipLaunchKernelGGL(k_sweep<1>, g, blk, 0, 0, C, P, ncp, B);
                else hipLaunchKernelGGL(k_sweep<2>, g, blk, 0, 0, C, P, ncp, B);
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("%-28s %.3f ms per sweep  %.2f TB/s\n", name, ms / 5, 2.0 * n * 8 / (ms / 5 * 1e-3) / 1e12);
        }
    };
    run(2, 1, "y pattern");
    run(0, 1, "z natural");
    for (int B : {4, 8, 16, 32, 64, 128}) { char nm[64]; snprintf(nm, 64, "z blocked B = %d", B); run(1, B, nm); }
    return 0;
}
