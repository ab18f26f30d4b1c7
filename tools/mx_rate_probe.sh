#!/bin/bash
# Timing-probe build for tools/mx_rate_probe.py: the product objects + gemm8.hip compiled with -DG8_MXPROBE=1 ->
# wsovod_amd/lib/abl/libmxprobe.so (the product library is never touched; delete lib/abl afterwards: it travels with every push).
set -e
cd "$(dirname "$0")/.."
python -c "from wsovod_amd import build; build.build()"
mkdir -p wsovod_amd/lib/abl
objs=$(ls wsovod_amd/csrc/build/*.o | grep -v "/gemm8.o")
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DG8_MXPROBE=1 -c wsovod_amd/csrc/gemm8.hip -o /tmp/gemm8_mxprobe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o wsovod_amd/lib/abl/libmxprobe.so /tmp/gemm8_mxprobe.o $objs
ls -la wsovod_amd/lib/abl
