// Probe: does padding the complex rows of the 512^3 half-complex grid (257 -> aligned) speed up rocFFT's strided passes?
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define FK(x) do { hipfftResult r = (x); if (r != HIPFFT_SUCCESS) { printf("hipfft error %d at %d\n", (int)r, __LINE__); exit(2); } } while (0)
int main(int argc, char** argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 512;
    const size_t maxC = (size_t)P * (P + 16) * 384;
    float2* C; float* rl;
    CK(hipMalloc(&C, maxC * 8)); CK(hipMalloc(&rl, (size_t)P * P * (P + 64) * 4));
    CK(hipMemset(C, 0, maxC * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int pads[] = {P / 2 + 1, P / 2 + 8, P / 2 + 16, P / 2 + 32, P / 2 + 64};
    for (int pi = 0; pi < 5; pi++) {
        const int nc = pads[pi];
        int n[3] = {P, P, P};
        int cE[3] = {P, P, nc}, rE[3] = {P, P, P};
        hipfftHandle c2r, r2c;
        FK(hipfftPlanMany(&c2r, 3, n, cE, 1, P * P * nc, rE, 1, P * P * P, HIPFFT_C2R, 1));
        FK(hipfftPlanMany(&r2c, 3, n, rE, 1, P * P * P, cE, 1, P * P * nc, HIPFFT_R2C, 1));
        float a = 0, b = 0, ms;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0)); FK(hipfftExecC2R(c2r, (hipfftComplex*)C, rl)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) a += ms / 3;
            CK(hipEventRecord(e0)); FK(hipfftExecR2C(r2c, rl, (hipfftComplex*)C)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) b += ms / 3;
        }
        printf("complex row %d: c2r %.3f ms  r2c %.3f ms\n", nc, a, b);
        hipfftDestroy(c2r); hipfftDestroy(r2c);
    }
    // plane padding as well: complex embed {P, n1e, n2e}
    {
        int n1s[] = {P, P + 1, P + 2, P + 4, P + 8, P + 16};
        int n2s[] = {P / 2 + 8, P / 2 + 16};
        for (int a2 = 0; a2 < 2; a2++)
            for (int a1 = 0; a1 < 6; a1++) {
                const int n1e = n1s[a1], nc = n2s[a2];
                if ((size_t)P * n1e * nc > maxC) continue;
                int n[3] = {P, P, P};
                int cE[3] = {P, n1e, nc}, rE[3] = {P, P, P};
                hipfftHandle c2r, r2c;
                FK(hipfftPlanMany(&c2r, 3, n, cE, 1, P * n1e * nc, rE, 1, P * P * P, HIPFFT_C2R, 1));
                FK(hipfftPlanMany(&r2c, 3, n, rE, 1, P * P * P, cE, 1, P * n1e * nc, HIPFFT_R2C, 1));
                float a = 0, b = 0, ms;
                for (int rep = 0; rep < 4; rep++) {
                    CK(hipEventRecord(e0)); FK(hipfftExecC2R(c2r, (hipfftComplex*)C, rl)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) a += ms / 3;
                    CK(hipEventRecord(e0)); FK(hipfftExecR2C(r2c, rl, (hipfftComplex*)C)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) b += ms / 3;
                }
                printf("complex embed {%d, %d, %d}: c2r %.3f ms  r2c %.3f ms\n", P, n1e, nc, a, b);
                hipfftDestroy(c2r); hipfftDestroy(r2c);
            }
    }
    // real side padded too: real embed {P, P, P + pad}
    {
        int pads[] = {8, 16, 32};
        for (int pi = 0; pi < 3; pi++) {
            const int nc = P / 2 + 8, nr = P + pads[pi];
            int n[3] = {P, P, P};
            int cE[3] = {P, P, nc}, rE[3] = {P, P, nr};
            hipfftHandle c2r, r2c;
            FK(hipfftPlanMany(&c2r, 3, n, cE, 1, P * P * nc, rE, 1, P * P * nr, HIPFFT_C2R, 1));
            FK(hipfftPlanMany(&r2c, 3, n, rE, 1, P * P * nr, cE, 1, P * P * nc, HIPFFT_R2C, 1));
            float a = 0, b = 0, ms;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0)); FK(hipfftExecC2R(c2r, (hipfftComplex*)C, rl)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) a += ms / 3;
                CK(hipEventRecord(e0)); FK(hipfftExecR2C(r2c, rl, (hipfftComplex*)C)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) b += ms / 3;
            }
            printf("real row %d (complex row %d): c2r %.3f ms  r2c %.3f ms\n", nr, nc, a, b);
            hipfftDestroy(c2r); hipfftDestroy(r2c);
        }
    }
    // in-place real layout (real rows padded to 2*(P/2+1))
    {
        const int nc = P / 2 + 1;
        int n[3] = {P, P, P};
        int cE[3] = {P, P, nc}, rE[3] = {P, P, 2 * nc};
        hipfftHandle c2r, r2c;
        FK(hipfftPlanMany(&c2r, 3, n, cE, 1, P * P * nc, rE, 1, P * P * 2 * nc, HIPFFT_C2R, 1));
        FK(hipfftPlanMany(&r2c, 3, n, rE, 1, P * P * 2 * nc, cE, 1, P * P * nc, HIPFFT_R2C, 1));
        float a = 0, b = 0, ms;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0)); FK(hipfftExecC2R(c2r, (hipfftComplex*)C, (float*)C)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) a += ms / 3;
            CK(hipEventRecord(e0)); FK(hipfftExecR2C(r2c, (float*)C, (hipfftComplex*)C)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) b += ms / 3;
        }
        printf("in-place: c2r %.3f ms  r2c %.3f ms\n", a, b);
    }
    return 0;
}
