#!/bin/bash
# builds the library (the prebuilt .so is what travels to the GPU box), checks that it exports the ABI, then hands the command to gpurun
# usage: tools/gpu.sh <timeout seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "
from thunder_amd import build, capi
build.build()
build.build_harness()
build.build_tools()
h = capi.load()
[getattr(h, n) for n in capi.SIGNATURES]
from oracle import oracle as O
O.build()
"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
