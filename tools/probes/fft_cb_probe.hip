// Probe: do hipFFT (rocFFT) load/store callbacks work on gfx950 for the 512^3 c2r / r2c of the balancing loop, and what
// do they cost?  Build: hipcc --offload-arch=gfx950 -O3 tools/fft_cb_probe.hip -lhipfft -o /tmp/fft_cb_probe
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <hipfft/hipfftXt.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define FK(x) do { hipfftResult r = (x); if (r != HIPFFT_SUCCESS) { printf("hipfft error %d at %d\n", (int)r, __LINE__); exit(2); } } while (0)

struct CbData { const float* T; const float* W; float* Wout; unsigned* diff; int P; float scale; };

__device__ hipfftComplex load_TW(void* in, size_t off, void* info, void*)
{
    const CbData* d = (const CbData*)info;
    const float w = d->W[off];
    hipfftComplex r; r.x = d->T[off] * w; r.y = 0.0f * w;
    return r;
}
__device__ void store_scale(void* out, size_t off, hipfftReal v, void* info, void*)
{
    const CbData* d = (const CbData*)info;
    const int P = d->P;
    const unsigned o = (unsigned)off;
    const int iw = o & (P - 1), jw = (o >> 9) & (P - 1), kw = o >> 18;   // P = 512
    const int i = iw >= P / 2 ? iw - P : iw, j = jw >= P / 2 ? jw - P : jw, k = kw >= P / 2 ? kw - P : kw;
    const float q = (float)(i * i + j * j + k * k);
    ((float*)out)[off] = v * d->scale * __expf(-q * 1e-4f);
}
__device__ void store_updW(void* out, size_t off, hipfftComplex c, void* info, void*)
{
    const CbData* d = (const CbData*)info;
    const float a = hypotf(c.x, c.y);
    const float m = a > 1e-6f ? a : 1e-6f;
    d->Wout[off] = d->W[off] / m;
    const unsigned bits = __float_as_uint(fabsf(a - 1.0f));
    if (bits > __hip_atomic_load(d->diff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(d->diff, bits);
}
__device__ hipfftCallbackLoadC p_load_TW = load_TW;
__device__ hipfftCallbackStoreR p_store_scale = store_scale;
__device__ hipfftCallbackStoreC p_store_updW = store_updW;

__global__ void k_calcC(float2* C, const float* T, const float* W, size_t n)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) { float w = W[e]; C[e] = make_float2(T[e] * w, 0.0f * w); }
}
__global__ void k_scale(float* rl, int P, float scale, size_t n)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const unsigned o = (unsigned)e;
    const int iw = o & (P - 1), jw = (o >> 9) & (P - 1), kw = o >> 18;
    const int i = iw >= P / 2 ? iw - P : iw, j = jw >= P / 2 ? jw - P : jw, k = kw >= P / 2 ? kw - P : kw;
    const float q = (float)(i * i + j * j + k * k);
    rl[e] = rl[e] * scale * __expf(-q * 1e-4f);
}
__global__ void k_updW(float* Wout, const float* W, const float2* C, unsigned* diff, size_t n)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float2 c = C[e];
    const float a = hypotf(c.x, c.y);
    const float m = a > 1e-6f ? a : 1e-6f;
    Wout[e] = W[e] / m;
    const unsigned bits = __float_as_uint(fabsf(a - 1.0f));
    if (bits > __hip_atomic_load(diff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(diff, bits);
}

int main()
{
    const int P = 512;
    const size_t nR = (size_t)P * P * P, nC = (size_t)P * P * (P / 2 + 1);
    float *T, *W, *W1, *W2, *rl1, *rl2;
    float2 *C1, *C2;
    unsigned* diff;
    CK(hipMalloc(&T, nC * 4)); CK(hipMalloc(&W, nC * 4)); CK(hipMalloc(&W1, nC * 4)); CK(hipMalloc(&W2, nC * 4));
    CK(hipMalloc(&rl1, nR * 4)); CK(hipMalloc(&rl2, nR * 4)); CK(hipMalloc(&C1, nC * 8)); CK(hipMalloc(&C2, nC * 8));
    CK(hipMalloc(&diff, 8)); CK(hipMemset(diff, 0, 8));
    std::vector<float> h(nC);
    srand(1);
    for (size_t i = 0; i < nC; i++) h[i] = (float)rand() / RAND_MAX + 0.5f;
    CK(hipMemcpy(T, h.data(), nC * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < nC; i++) h[i] = (float)rand() / RAND_MAX + 0.5f;
    CK(hipMemcpy(W, h.data(), nC * 4, hipMemcpyHostToDevice));
    CbData hd = {T, W, W2, diff + 1, P, 1.0f / (float)nR};
    CbData* dd; CK(hipMalloc(&dd, sizeof(CbData))); CK(hipMemcpy(dd, &hd, sizeof(hd), hipMemcpyHostToDevice));

    hipfftHandle c2r, r2c, c2rCb, r2cCb;
    FK(hipfftPlan3d(&c2r, P, P, P, HIPFFT_C2R)); FK(hipfftPlan3d(&r2c, P, P, P, HIPFFT_R2C));
    FK(hipfftPlan3d(&c2rCb, P, P, P, HIPFFT_C2R)); FK(hipfftPlan3d(&r2cCb, P, P, P, HIPFFT_R2C));
    void *fLoad, *fStoreR, *fStoreC;
    CK(hipMemcpyFromSymbol(&fLoad, HIP_SYMBOL(p_load_TW), sizeof(void*)));
    CK(hipMemcpyFromSymbol(&fStoreR, HIP_SYMBOL(p_store_scale), sizeof(void*)));
    CK(hipMemcpyFromSymbol(&fStoreC, HIP_SYMBOL(p_store_updW), sizeof(void*)));
    void* cbd = dd;
    hipfftResult r;
    r = hipfftXtSetCallback(c2rCb, &fLoad, HIPFFT_CB_LD_COMPLEX, &cbd); printf("set load cb: %d\n", (int)r);
    if (r != HIPFFT_SUCCESS) return 3;
    r = hipfftXtSetCallback(c2rCb, &fStoreR, HIPFFT_CB_ST_REAL, &cbd); printf("set store-real cb: %d\n", (int)r);
    if (r != HIPFFT_SUCCESS) return 3;
    r = hipfftXtSetCallback(r2cCb, &fStoreC, HIPFFT_CB_ST_COMPLEX, &cbd); printf("set store-complex cb: %d\n", (int)r);
    if (r != HIPFFT_SUCCESS) return 3;

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nb = (int)((nC + 255) / 256), nbR = (int)((nR + 255) / 256);
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        // plain path
        CK(hipMemset(diff, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_calcC, dim3(nb), dim3(256), 0, 0, C1, T, W, nC);
        FK(hipfftExecC2R(c2r, (hipfftComplex*)C1, rl1));
        hipLaunchKernelGGL(k_scale, dim3(nbR), dim3(256), 0, 0, rl1, P, 1.0f / (float)nR, nR);
        FK(hipfftExecR2C(r2c, rl1, (hipfftComplex*)C1));
        hipLaunchKernelGGL(k_updW, dim3(nb), dim3(256), 0, 0, W1, W, C1, diff, nC);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("plain round: %.3f ms\n", ms);
        // callback path
        CK(hipEventRecord(e0));
        FK(hipfftExecC2R(c2rCb, (hipfftComplex*)C2, rl2));
        FK(hipfftExecR2C(r2cCb, rl2, (hipfftComplex*)C2));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("callback round: %.3f ms\n", ms);
    }
    // which callback costs what: one at a time on fresh plans
    {
        hipfftHandle pa, pb, pc;
        FK(hipfftPlan3d(&pa, P, P, P, HIPFFT_C2R)); FK(hipfftPlan3d(&pb, P, P, P, HIPFFT_C2R)); FK(hipfftPlan3d(&pc, P, P, P, HIPFFT_R2C));
        FK(hipfftXtSetCallback(pa, &fLoad, HIPFFT_CB_LD_COMPLEX, &cbd));
        FK(hipfftXtSetCallback(pb, &fStoreR, HIPFFT_CB_ST_REAL, &cbd));
        FK(hipfftXtSetCallback(pc, &fStoreC, HIPFFT_CB_ST_COMPLEX, &cbd));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0)); FK(hipfftExecC2R(c2r, (hipfftComplex*)C1, rl1)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("c2r plain %.3f", ms);
            CK(hipEventRecord(e0)); FK(hipfftExecC2R(pa, (hipfftComplex*)C2, rl2)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  c2r load-cb %.3f", ms);
            CK(hipEventRecord(e0)); FK(hipfftExecC2R(pb, (hipfftComplex*)C1, rl2)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  c2r store-cb %.3f", ms);
            CK(hipEventRecord(e0)); FK(hipfftExecR2C(r2c, rl1, (hipfftComplex*)C1)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  r2c plain %.3f", ms);
            CK(hipEventRecord(e0)); FK(hipfftExecR2C(pc, rl1, (hipfftComplex*)C2)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  r2c store-cb %.3f\n", ms);
        }
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_calcC, dim3(nb), dim3(256), 0, 0, C1, T, W, nC); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_calcC %.3f", ms);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_scale, dim3(nbR), dim3(256), 0, 0, rl1, P, 1.0f / (float)nR, nR); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  k_scale %.3f", ms);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_updW, dim3(nb), dim3(256), 0, 0, W1, W, C1, diff, nC); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  k_updW %.3f\n", ms);
    }
    // compare
    std::vector<float> a(nC), b(nC);
    CK(hipMemcpy(a.data(), W1, nC * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), W2, nC * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < nC; i++) { md = fmax(md, fabs((double)a[i] - b[i])); mx = fmax(mx, fabs((double)a[i])); }
    unsigned hb[2]; CK(hipMemcpy(hb, diff, 8, hipMemcpyDeviceToHost));
    float d0, d1; memcpy(&d0, &hb[0], 4); memcpy(&d1, &hb[1], 4);
    printf("max |W1-W2| = %g (max |W1| %g); diff plain %g cb %g\n", md, mx, d0, d1);
    return 0;
}
