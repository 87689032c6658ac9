# builds thunder_amd/lib/libthunder_amd_alt.so with extra compile flags, for A/B runs through THX_LIB:
#   bash tools/build_alt.sh "-DTHX_ACC_STRIDED=0" [name: libthunder_amd_<name>.so, default alt]
set -e
cd "$(dirname "$0")/.."
NAME=${2:-alt}; D=thunder_amd/lib/$NAME.d; mkdir -p $D
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result $1"
for s in thunder_amd/csrc/*.hip; do /opt/rocm/bin/hipcc $FLAGS -c $s -o $D/$(basename $s .hip).o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o thunder_amd/lib/libthunder_amd_$NAME.so $D/*.o -lhipfft -lrccl
rm -rf $D; ls -la thunder_amd/lib/libthunder_amd_$NAME.so
