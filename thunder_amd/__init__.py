"""thunder_amd -- MI355X-native E/M hot path of thuem/THUNDER (Projector slice extraction + CTF,
particle-filter likelihood/weights, Reconstructor Fourier insertion + Wiener/gridding reconstruction)
behind the reference's Projector / Reconstructor / Interface.h surfaces.  See DESIGN.md.

The product path is the HIP library thunder_amd/lib/libthunder_amd.so (C ABI: include/thunder_amd.h);
there is no CPU fallback.
"""
__version__ = "0.1.0"
