// thx_philox.h -- counter-based random numbers of the device-side particle filter and of the insertion draws.
// The reference draws from GSL's global mt19937 under OpenMP (unordered, not reproducible); here every draw is a pure
// function of (seed, image, call, purpose, index), independent of launch geometry.  tests/_philox.py is the numpy replica.
#pragma once
#include <hip/hip_runtime.h>

namespace thx {

// ---- Philox4x32-10 (Salmon et al., SC'11) ----
struct Philox {
    unsigned k0, k1;
    __host__ __device__ __forceinline__ void round(unsigned c[4], unsigned ka, unsigned kb) const
    {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ ka, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ kb, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __host__ __device__ __forceinline__ void operator()(unsigned c[4]) const
    {
        unsigned ka = k0, kb = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            round(c, ka, kb);
            ka += 0x9E3779B9u;
            kb += 0xBB67AE85u;
        }
    }
};

// four uniforms in (0, 1) / four standard normals for (image, call, purpose, index)
__host__ __device__ __forceinline__ void draw_u4(double u[4], unsigned long long seed, unsigned img, unsigned call, unsigned purpose,
                                        unsigned index)
{
    Philox g{(unsigned)seed, (unsigned)(seed >> 32)};
    unsigned c[4] = {img, call, purpose, index};
    g(c);
#pragma unroll
    for (int i = 0; i < 4; i++) u[i] = ((double)c[i] + 0.5) * (1.0 / 4294967296.0);
}
__device__ __forceinline__ void draw_n4(double n[4], unsigned long long seed, unsigned img, unsigned call, unsigned purpose,
                                        unsigned index)
{
    double u[4];
    draw_u4(u, seed, img, call, purpose, index);
    const double r0 = sqrt(-2.0 * log(u[0])), r1 = sqrt(-2.0 * log(u[2]));
    double s, c;
    sincos(6.283185307179586476925 * u[1], &s, &c);
    n[0] = r0 * c; n[1] = r0 * s;
    sincos(6.283185307179586476925 * u[3], &s, &c);
    n[2] = r1 * c; n[3] = r1 * s;
}

}  // namespace thx
