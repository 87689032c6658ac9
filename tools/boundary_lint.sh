#!/bin/bash
# Boundary lint (NOT parity evidence, and NOT a build of the reference): `g++ -fsyntax-only` of the reference-side replacement file
# integration/Interface_thx.cpp against the reference's OWN, unchanged headers -- gpu/interface/Interface.h and everything it pulls in
# (Volume.h, Image.h, Particle.h, Database.h, ...) -- so that a compiler, not a regular expression, checks the prototypes, the
# container accessors (&F3D[0], kernelRL.getData(), mgr->handle()) and the casts.  Nothing is linked and nothing runs.
# What the image lacks is generated as THROW-AWAY stand-ins under /tmp (never committed, never used by any build or test):
#   gsl/      symlinks to the gsl_*.h headers of the vendored GSL 2.4 source tree (its `make` would create the same directory)
#   THUNDERConfig.h   the cmake-generated configuration header (SINGLE_PRECISION, GPU_VERSION, ENABLE_SIMD_256)
#   boost/    empty-bodied stand-ins for the five boost headers the reference's headers include (boost 1.60 is a missing blob)
#   mpi.h     only if the image has none
# The two replacement headers integration/Managed{ArrayTexture,CalPoint}.h are pre-included: in the reference's tree they REPLACE
# gpu/include/Managed*.h (same include guards), which gpu/include/cuthunder.h would otherwise find next to itself.
# usage: tools/boundary_lint.sh [reference root]   (exit code = g++'s)
set -u
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")/.." && pwd)
S=/tmp/thx_lint
rm -rf $S; mkdir -p $S/gsl $S/boost/move $S/boost/container
find $REF/external/packages/gsl-2.4 -name 'gsl_*.h' | while read f; do ln -sf "$f" $S/gsl/$(basename "$f"); done
[ -f $S/gsl/gsl_version.h ] || cat > $S/gsl/gsl_version.h <<'H'
#define GSL_VERSION "2.4"
#define GSL_MAJOR_VERSION 2
#define GSL_MINOR_VERSION 4
H
cat > $S/THUNDERConfig.h <<'H'
#define THUNDER_VERSION_MAJOR 1
#define THUNDER_VERSION_MINOR 4
#define THUNDER_VERSION_ADDIT 14
#define COMMIT_VERSION_QUOTE "lint"
#define SINGLE_PRECISION
#ifndef GPU_VERSION
#define GPU_VERSION
#endif
#define ENABLE_SIMD_256
H
# boost: just enough declarations for the reference's headers to parse
cat > $S/boost/noncopyable.hpp <<'H'
#pragma once
namespace boost { class noncopyable { protected: noncopyable() {} ~noncopyable() {} private: noncopyable(const noncopyable&); noncopyable& operator=(const noncopyable&); }; }
H
cat > $S/boost/move/core.hpp <<'H'
#pragma once
#include <utility>
#define BOOST_MOVABLE_BUT_NOT_COPYABLE(T) private: T(const T&); T& operator=(const T&); public:
#define BOOST_COPYABLE_AND_MOVABLE(T)
#define BOOST_RV_REF(T) T&&
#define BOOST_COPY_ASSIGN_REF(T) const T&
#define BOOST_FWD_REF(T) T&&
#define BOOST_MOVE_BASE(B, x) static_cast<B&&>(x)
namespace boost { template <class T> T&& move(T& t) { return static_cast<T&&>(t); } template <class T> T&& move(T&& t) { return static_cast<T&&>(t); } }
H
cat > $S/boost/move/make_unique.hpp <<'H'
#pragma once
#include <memory>
namespace boost { namespace movelib {
template <class T> struct unique_ptr : std::unique_ptr<T> { using std::unique_ptr<T>::unique_ptr; };
template <class T, class... A> std::unique_ptr<T> make_unique(A&&... a) { return std::unique_ptr<T>(new T(static_cast<A&&>(a)...)); }
} }
H
cat > $S/boost/move/unique_ptr.hpp <<'H'
#pragma once
#include "make_unique.hpp"
H
cat > $S/boost/container/vector.hpp <<'H'
#pragma once
#include <vector>
namespace boost { namespace container { template <class T> class vector : public std::vector<T> { public: using std::vector<T>::vector; }; } }
H
cat > $S/boost/function.hpp <<'H'
#pragma once
#include <functional>
namespace boost { template <class S> class function : public std::function<S> { public: using std::function<S>::function; }; }
H
cat > $S/boost/bind.hpp <<'H'
#pragma once
#include <functional>
namespace boost { using std::bind; using std::ref; using std::cref; namespace placeholders { using namespace std::placeholders; } }
using namespace std::placeholders;
H
MPI_INC=""
if ! echo '#include <mpi.h>' | g++ -x c++ -fsyntax-only - 2>/dev/null; then
    for d in /opt/conda/include /usr/include/x86_64-linux-gnu/mpi /usr/lib/x86_64-linux-gnu/openmpi/include; do
        [ -f $d/mpi.h ] && { mkdir -p $S/mpi; ln -sf $d/mpi.h $S/mpi/mpi.h; for f in $d/mpi*.h $d/mpio.h; do [ -f $f ] && ln -sf $f $S/mpi/; done; MPI_INC="-I$S/mpi"; break; }
    done
fi
STD=${LINT_STD:-gnu++11}
g++ -std=$STD -fsyntax-only -fopenmp -mavx2 -mfma -Wall -Wno-unknown-pragmas -Wno-ignored-attributes -DGPU_VERSION \
    -include $HERE/integration/ManagedArrayTexture.h -include $HERE/integration/ManagedCalPoint.h \
    -I$HERE/integration -I$S $MPI_INC -I$HERE/include \
    -I$REF/gpu/interface -I$REF/include -I$REF/include/Functions -I$REF/include/Geometry -I$REF/include/Image \
    -I$REF/gpu/include -I$REF/external/Eigen3 -I$REF/external/easylogging -I$REF/external/jsoncpp \
    -I$REF/external/packages/fftw-3.3.7/api \
    $HERE/integration/Interface_thx.cpp > $S/interface.log 2>&1
rc=$?
grep -A3 "^$HERE/.*\(warning\|error\)" $S/interface.log    # diagnostics in THIS repository's files (the reference's own headers warn plenty)
[ $rc -ne 0 ] && grep -B2 -A6 "error" $S/interface.log | head -80
echo "boundary lint: g++ -std=$STD -fsyntax-only integration/Interface_thx.cpp against $REF headers: exit $rc"
# the CPU build's call sites, in the reference's own types, against the class mirrors
g++ -std=$STD -fsyntax-only -fopenmp -mavx2 -mfma -Wall -Wno-unknown-pragmas -Wno-ignored-attributes \
    -I$S $MPI_INC -I$HERE/include -I$REF/include -I$REF/include/Functions -I$REF/include/Geometry -I$REF/include/Image \
    -I$REF/external/Eigen3 -I$REF/external/easylogging -I$REF/external/jsoncpp -I$REF/external/packages/fftw-3.3.7/api \
    $HERE/integration/callsite_lint.cpp > $S/callsite.log 2>&1
rc2=$?
grep -A3 "^$HERE/.*\(warning\|error\)" $S/callsite.log
[ $rc2 -ne 0 ] && grep -B2 -A6 "error" $S/callsite.log | head -80
echo "boundary lint: g++ -std=$STD -fsyntax-only integration/callsite_lint.cpp (reference call syntax against include/thunder_amd/*.hpp): exit $rc2"
# the reference's OWN, UNCHANGED callers of the plug-in surface -- src/Optimiser.cpp (ExpectLocal* / ExpectGlobal3D / ManagedArrayTexture /
# ManagedCalPoint, :2180-3393) and src/Reconstructor.cpp (InsertFT / PrepareTF / Expose*, :865-2330) -- with -DGPU_VERSION against the
# same header set, i.e. with gpu/include/Managed*.h replaced by integration/Managed*.h: they must still parse, since they are what
# links against integration/Interface_thx.cpp
rc3=0
for unit in Optimiser Reconstructor; do
    g++ -std=$STD -fsyntax-only -fopenmp -mavx2 -mfma -w -DGPU_VERSION \
        -include $HERE/integration/ManagedArrayTexture.h -include $HERE/integration/ManagedCalPoint.h \
        -I$HERE/integration -I$S $MPI_INC -I$HERE/include \
        -I$REF/gpu/interface -I$REF/include -I$REF/include/Functions -I$REF/include/Geometry -I$REF/include/Image \
        -I$REF/gpu/include -I$REF/external/Eigen3 -I$REF/external/easylogging -I$REF/external/jsoncpp \
        -I$REF/external/packages/fftw-3.3.7/api $REF/src/$unit.cpp > $S/$unit.log 2>&1
    r=$?
    [ $r -ne 0 ] && { rc3=$r; grep -B2 -A6 "error" $S/$unit.log | head -60; }
    echo "boundary lint: g++ -std=$STD -fsyntax-only -DGPU_VERSION <reference>/src/$unit.cpp (unchanged) with the replaced Managed*.h: exit $r"
done
[ $rc -eq 0 ] && [ $rc2 -eq 0 ] && [ $rc3 -eq 0 ]
