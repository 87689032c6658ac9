// Probe: the balancing loop's Fourier-space array C = T*W is REAL, so its 3-D inverse transform can be done as
// [1-D r2c along z, strided] + [P/2+1 batched 2-D c2r planes] on half the data.  How fast is that with rocFFT?
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define FK(x) do { hipfftResult r = (x); if (r != HIPFFT_SUCCESS) { printf("hipfft error %d at %d\n", (int)r, __LINE__); exit(2); } } while (0)
static float timeit(hipEvent_t e0, hipEvent_t e1) { float ms; CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); return ms; }
int main(int argc, char** argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 512;
    const int nh = P / 2 + 1;
    for (int ncp = nh; ncp <= nh + 7; ncp += 7) {   // 257 and 264
        const int plane = P * ncp;                   // elements per kz plane
        float* R; float2* G; float* g;
        CK(hipMalloc(&R, (size_t)P * plane * 4));            // real Fourier array [kz][ky][kx]
        CK(hipMalloc(&G, (size_t)nh * plane * 8));           // [z half][ky][kx] complex
        CK(hipMalloc(&g, (size_t)nh * P * P * 4));           // [z half][y][x] real
        CK(hipMemset(R, 0, (size_t)P * plane * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        // 1-D along z: n = P, real stride = plane, dist = 1, batch = plane
        hipfftHandle zr2c, zc2r, p2c2r, p2r2c;
        int n1[1] = {P};
        int e1d[1] = {P};
        hipfftResult r;
        r = hipfftPlanMany(&zr2c, 1, n1, e1d, plane, 1, e1d, plane, 1, HIPFFT_R2C, plane);
        printf("ncp %d: plan z r2c: %d\n", ncp, (int)r);
        if (r != HIPFFT_SUCCESS) continue;
        r = hipfftPlanMany(&zc2r, 1, n1, e1d, plane, 1, e1d, plane, 1, HIPFFT_C2R, plane);
        printf("plan z c2r: %d\n", (int)r);
        if (r != HIPFFT_SUCCESS) continue;
        int n2[2] = {P, P};
        int cE[2] = {P, ncp}, rE[2] = {P, P};
        FK(hipfftPlanMany(&p2c2r, 2, n2, cE, 1, plane, rE, 1, P * P, HIPFFT_C2R, nh));
        FK(hipfftPlanMany(&p2r2c, 2, n2, rE, 1, P * P, cE, 1, plane, HIPFFT_R2C, nh));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0)); FK(hipfftExecR2C(zr2c, R, (hipfftComplex*)G)); CK(hipEventRecord(e1)); float a = timeit(e0, e1);
            CK(hipEventRecord(e0)); FK(hipfftExecC2R(p2c2r, (hipfftComplex*)G, g)); CK(hipEventRecord(e1)); float b = timeit(e0, e1);
            CK(hipEventRecord(e0)); FK(hipfftExecR2C(p2r2c, g, (hipfftComplex*)G)); CK(hipEventRecord(e1)); float c = timeit(e0, e1);
            CK(hipEventRecord(e0)); FK(hipfftExecC2R(zc2r, (hipfftComplex*)G, R)); CK(hipEventRecord(e1)); float d = timeit(e0, e1);
            printf("  z r2c %.3f  planes c2r %.3f | planes r2c %.3f  z c2r %.3f   (inverse %.3f, forward %.3f ms)\n", a, b, c, d, a + b, c + d);
        }
        if (ncp % 2 == 0) {
            // pairs of real columns as one complex column: 1-D c2c along z, stride = batch = P * ncp / 2
            hipfftHandle zc2c;
            const int nb = plane / 2;
            r = hipfftPlanMany(&zc2c, 1, n1, e1d, nb, 1, e1d, nb, 1, HIPFFT_C2C, nb);
            printf("plan z c2c (paired columns): %d\n", (int)r);
            if (r == HIPFFT_SUCCESS) {
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0)); FK(hipfftExecC2C(zc2c, (hipfftComplex*)R, (hipfftComplex*)R, HIPFFT_BACKWARD)); CK(hipEventRecord(e1)); float a = timeit(e0, e1);
                    CK(hipEventRecord(e0)); FK(hipfftExecC2C(zc2c, (hipfftComplex*)R, (hipfftComplex*)R, HIPFFT_FORWARD)); CK(hipEventRecord(e1)); float b = timeit(e0, e1);
                    printf("  z c2c in place: backward %.3f  forward %.3f ms\n", a, b);
                }
                hipfftDestroy(zc2c);
            }
        }
        if (ncp % 2 == 0) {
            // 2-D c2c over the two slow dimensions (z, y) with the paired kx columns as the (fastest, unit-distance) batch
            hipfftHandle zy;
            const int nb = ncp / 2;
            int n2d[2] = {P, P}, e2d[2] = {P, P};
            r = hipfftPlanMany(&zy, 2, n2d, e2d, nb, 1, e2d, nb, 1, HIPFFT_C2C, nb);
            printf("plan (z,y) c2c, batch %d: %d\n", nb, (int)r);
            if (r == HIPFFT_SUCCESS) {
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0)); FK(hipfftExecC2C(zy, (hipfftComplex*)R, (hipfftComplex*)R, HIPFFT_BACKWARD)); CK(hipEventRecord(e1)); float a = timeit(e0, e1);
                    CK(hipEventRecord(e0)); FK(hipfftExecC2C(zy, (hipfftComplex*)R, (hipfftComplex*)R, HIPFFT_FORWARD)); CK(hipEventRecord(e1)); float b = timeit(e0, e1);
                    printf("  (z,y) c2c in place: backward %.3f  forward %.3f ms\n", a, b);
                }
                hipfftDestroy(zy);
            }
            // contiguous rows: 1-D c2r / r2c along x for the half volume
            hipfftHandle xr, xf;
            int cEx[1] = {ncp}, rEx[1] = {P};
            FK(hipfftPlanMany(&xr, 1, n1, cEx, 1, ncp, rEx, 1, P, HIPFFT_C2R, nh * P));
            FK(hipfftPlanMany(&xf, 1, n1, rEx, 1, P, cEx, 1, ncp, HIPFFT_R2C, nh * P));
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0)); FK(hipfftExecC2R(xr, (hipfftComplex*)G, g)); CK(hipEventRecord(e1)); float a = timeit(e0, e1);
                CK(hipEventRecord(e0)); FK(hipfftExecR2C(xf, g, (hipfftComplex*)G)); CK(hipEventRecord(e1)); float b = timeit(e0, e1);
                printf("  x rows half volume: c2r %.3f  r2c %.3f ms\n", a, b);
            }
            hipfftDestroy(xr); hipfftDestroy(xf);
        }
        hipfftDestroy(zr2c); hipfftDestroy(zc2r); hipfftDestroy(p2c2r); hipfftDestroy(p2r2c);
        hipFree(R); hipFree(G); hipFree(g);
    }
    return 0;
}
