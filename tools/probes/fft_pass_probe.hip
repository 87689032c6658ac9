// Probe: a hand-written strided 512-point FFT pass (radix 8 x 8 x 8, one LDS tile of 512 x TX complex per workgroup)
// against rocFFT's strided passes on the padded [512][512][264] half-complex grid of the gridding iteration.
// Checks the full 3-D r2c / c2r chain (rocFFT 1-D batched along x + two hand-written passes) against the 3-D rocFFT plan.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define FK(x) do { hipfftResult r = (x); if (r != HIPFFT_SUCCESS) { printf("hipfft error %d at %d\n", (int)r, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
template <int DIR> __device__ __forceinline__ float2 mul_i(float2 a) { return DIR > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }

// y[k] = sum_n v[n] exp(DIR 2 pi i n k / 8), in place (decimation in frequency, outputs in natural order)
template <int DIR>
__device__ __forceinline__ void fft8(float2 v[8])
{
    const float h = 0.70710678118654752f;
    float2 s0 = cadd(v[0], v[4]), s1 = cadd(v[1], v[5]), s2 = cadd(v[2], v[6]), s3 = cadd(v[3], v[7]);
    float2 d0 = csub(v[0], v[4]), d1 = csub(v[1], v[5]), d2 = csub(v[2], v[6]), d3 = csub(v[3], v[7]);
    // d[n] *= w8^n
    {
        const float2 t1 = mul_i<DIR>(d1);   // d1 * (DIR i)
        d1 = make_float2((d1.x + t1.x) * h, (d1.y + t1.y) * h);          // (1 + DIR i) / sqrt 2
        d2 = mul_i<DIR>(d2);
        const float2 t3 = mul_i<DIR>(d3);
        d3 = make_float2((t3.x - d3.x) * h, (t3.y - d3.y) * h);          // (-1 + DIR i) / sqrt 2
    }
    // 4-point DFTs
    {
        const float2 e0 = cadd(s0, s2), e1 = csub(s0, s2), o0 = cadd(s1, s3), o1 = mul_i<DIR>(csub(s1, s3));
        v[0] = cadd(e0, o0); v[4] = csub(e0, o0); v[2] = cadd(e1, o1); v[6] = csub(e1, o1);
    }
    {
        const float2 e0 = cadd(d0, d2), e1 = csub(d0, d2), o0 = cadd(d1, d3), o1 = mul_i<DIR>(csub(d1, d3));
        v[1] = cadd(e0, o0); v[5] = csub(e0, o0); v[3] = cadd(e1, o1); v[7] = csub(e1, o1);
    }
}

// One workgroup: 512 points x TX adjacent columns, in place.  data[b * strideB + e * strideE + x], x = x0 + c.
template <int TX, int DIR>
__global__ __launch_bounds__(64 * TX) void k_fft512_strided(float2* __restrict__ data, long strideE, long strideB, int nx,
                                                           const float2* __restrict__ tw)
{
    extern __shared__ float2 lds[];          // (512 + 8) rows x TX, then 512 twiddles
    float2* sTw = lds + 520 * TX;
    const int c = threadIdx.x % TX, t = threadIdx.x / TX;
    const int x = blockIdx.x * TX + c;
    const bool ok = x < nx;
    float2* base = data + (long)blockIdx.y * strideB + x;
    for (int i = threadIdx.x; i < 512; i += 64 * TX) sTw[i] = tw[i];
    float2 v[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; n2++) v[n2] = ok ? base[(long)(t + 64 * n2) * strideE] : make_float2(0.f, 0.f);
    fft8<DIR>(v);
    __syncthreads();
    {
        const int n1 = t >> 3;
#pragma unroll
        for (int k0 = 0; k0 < 8; k0++) {
            const float2 w = sTw[(8 * n1 * k0) & 511];
            lds[(k0 * 65 + t) * TX + c] = cmul(v[k0], w);
        }
    }
    __syncthreads();
    {
        const int n0 = t & 7, k0 = t >> 3;
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) v[n1] = lds[(k0 * 65 + n1 * 8 + n0) * TX + c];
        fft8<DIR>(v);
#pragma unroll
        for (int k1 = 0; k1 < 8; k1++) {
            const float2 w = sTw[n0 * (k0 + 8 * k1)];
            lds[(k0 * 65 + k1 * 8 + n0) * TX + c] = cmul(v[k1], w);
        }
    }
    __syncthreads();
    {
        const int k0 = t & 7, k1 = t >> 3;
#pragma unroll
        for (int n0 = 0; n0 < 8; n0++) v[n0] = lds[(k0 * 65 + k1 * 8 + n0) * TX + c];
        fft8<DIR>(v);
        if (ok) {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) base[(long)(t + 64 * k2) * strideE] = v[k2];
        }
    }
}

template <int TX, int DIR>
static void launch_pass(float2* data, long strideE, long strideB, int nBatch, int nx, const float2* tw, hipStream_t st)
{
    const size_t lds = (size_t)(520 * TX + 512) * sizeof(float2);
    static bool done = false;
    if (!done) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fft512_strided<TX, DIR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        done = true;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fft512_strided<TX, DIR>), dim3((nx + TX - 1) / TX, nBatch), dim3(64 * TX), lds, st, data, strideE,
                       strideB, nx, tw);
}

template <typename F>
static float time_ms(F f, int reps = 20)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    const int P = 512, ncp = 264;
    const size_t nC = (size_t)P * P * ncp, nR = (size_t)P * P * P;
    float2 *C, *C2, *twF, *twB; float *rl, *rl2;
    CK(hipMalloc(&C, nC * 8)); CK(hipMalloc(&C2, nC * 8)); CK(hipMalloc(&rl, nR * 4)); CK(hipMalloc(&rl2, nR * 4));
    CK(hipMalloc(&twF, 512 * 8)); CK(hipMalloc(&twB, 512 * 8));
    {
        std::vector<float2> f(512), b(512);
        for (int m = 0; m < 512; m++) {
            const double a = 2.0 * M_PI * m / 512.0;
            f[m] = make_float2((float)cos(a), (float)-sin(a));
            b[m] = make_float2((float)cos(a), (float)sin(a));
        }
        CK(hipMemcpy(twF, f.data(), 512 * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(twB, b.data(), 512 * 8, hipMemcpyHostToDevice));
        std::vector<float> h(nR);
        unsigned s = 12345u;
        for (size_t i = 0; i < nR; i++) { s = s * 1664525u + 1013904223u; h[i] = (float)((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
        CK(hipMemcpy(rl, h.data(), nR * 4, hipMemcpyHostToDevice));
    }
    int n3[3] = {P, P, P}, cE[3] = {P, P, ncp}, rE[3] = {P, P, P};
    hipfftHandle r2c3, c2r3, r2c1, c2r1;
    FK(hipfftPlanMany(&r2c3, 3, n3, rE, 1, P * P * P, cE, 1, P * P * ncp, HIPFFT_R2C, 1));
    FK(hipfftPlanMany(&c2r3, 3, n3, cE, 1, P * P * ncp, rE, 1, P * P * P, HIPFFT_C2R, 1));
    int n1[1] = {P}, e1r[1] = {P}, e1c[1] = {ncp};
    FK(hipfftPlanMany(&r2c1, 1, n1, e1r, 1, P, e1c, 1, ncp, HIPFFT_R2C, P * P));
    FK(hipfftPlanMany(&c2r1, 1, n1, e1c, 1, ncp, e1r, 1, P, HIPFFT_C2R, P * P));
    CK(hipMemset(C, 0, nC * 8)); CK(hipMemset(C2, 0, nC * 8));

    // ---- forward: reference 3-D plan vs x (rocFFT) + y + z (hand-written) ----
    FK(hipfftExecR2C(r2c3, rl, (hipfftComplex*)C));
    auto fwd8 = [&]() {
        FK(hipfftExecR2C(r2c1, rl, (hipfftComplex*)C2));
        launch_pass<8, -1>(C2, ncp, (long)P * ncp, P, ncp, twF, nullptr);
        launch_pass<8, -1>(C2, (long)P * ncp, ncp, P, ncp, twF, nullptr);
    };
    fwd8();
    CK(hipDeviceSynchronize());
    {
        std::vector<float2> a(nC), b(nC);
        CK(hipMemcpy(a.data(), C, nC * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), C2, nC * 8, hipMemcpyDeviceToHost));
        double dmax = 0, amax = 0;
        for (size_t r = 0; r < (size_t)P * P; r++)
            for (int i = 0; i < P / 2 + 1; i++) {
                const float2 u = a[r * ncp + i], w = b[r * ncp + i];
                dmax = fmax(dmax, fmax(fabs((double)u.x - w.x), fabs((double)u.y - w.y)));
                amax = fmax(amax, fmax(fabs((double)u.x), fabs((double)u.y)));
            }
        printf("forward: max |diff| %.3g of max %.3g (rel %.2e)\n", dmax, amax, dmax / amax);
    }
    // ---- backward ----
    FK(hipfftExecC2R(c2r3, (hipfftComplex*)C, rl2));   // note: c2r may overwrite its input; C is rebuilt below
    FK(hipfftExecR2C(r2c3, rl, (hipfftComplex*)C));
    CK(hipMemcpy(C2, C, nC * 8, hipMemcpyDeviceToDevice));
    float* rl3; CK(hipMalloc(&rl3, nR * 4));
    launch_pass<8, 1>(C2, (long)P * ncp, ncp, P, ncp, twB, nullptr);
    launch_pass<8, 1>(C2, ncp, (long)P * ncp, P, ncp, twB, nullptr);
    FK(hipfftExecC2R(c2r1, (hipfftComplex*)C2, rl3));
    CK(hipDeviceSynchronize());
    {
        std::vector<float> a(nR), b(nR);
        CK(hipMemcpy(a.data(), rl2, nR * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), rl3, nR * 4, hipMemcpyDeviceToHost));
        double dmax = 0, amax = 0;
        for (size_t i = 0; i < nR; i++) { dmax = fmax(dmax, fabs((double)a[i] - b[i])); amax = fmax(amax, fabs((double)a[i])); }
        printf("backward: max |diff| %.3g of max %.3g (rel %.2e)\n", dmax, amax, dmax / amax);
    }
    // ---- timing ----
    printf("rocFFT 3-D r2c %.3f ms, c2r %.3f ms\n", time_ms([&]() { FK(hipfftExecR2C(r2c3, rl, (hipfftComplex*)C)); }),
           time_ms([&]() { FK(hipfftExecC2R(c2r3, (hipfftComplex*)C, rl2)); }));
    printf("rocFFT 1-D batched x: r2c %.3f ms, c2r %.3f ms\n", time_ms([&]() { FK(hipfftExecR2C(r2c1, rl, (hipfftComplex*)C2)); }),
           time_ms([&]() { FK(hipfftExecC2R(c2r1, (hipfftComplex*)C2, rl3)); }));
    printf("hand y-pass TX=8 %.3f ms, TX=16 %.3f ms\n",
           time_ms([&]() { launch_pass<8, -1>(C2, ncp, (long)P * ncp, P, ncp, twF, nullptr); }),
           time_ms([&]() { launch_pass<16, -1>(C2, ncp, (long)P * ncp, P, ncp, twF, nullptr); }));
    printf("hand z-pass TX=8 %.3f ms, TX=16 %.3f ms\n",
           time_ms([&]() { launch_pass<8, -1>(C2, (long)P * ncp, ncp, P, ncp, twF, nullptr); }),
           time_ms([&]() { launch_pass<16, -1>(C2, (long)P * ncp, ncp, P, ncp, twF, nullptr); }));
    printf("hand y-pass over 257 columns only: TX=8 %.3f ms\n",
           time_ms([&]() { launch_pass<8, -1>(C2, ncp, (long)P * ncp, P, 257, twF, nullptr); }));
    printf("full hand chain r2c (x + y + z) %.3f ms\n", time_ms(fwd8));
    return 0;
}
