#!/bin/bash
# Ablation builds of the persistent 64-channel 3x3 kernel: links wsovod_amd/lib/abl/lib<N>.so for every C64P_ABL value
# given (see gemm.hip), from the product objects plus a gemm.hip compiled with -DC64P_ABL=<N>.  Run on the GPU box as
#   WSOVOD_LIB=$PWD/wsovod_amd/lib/abl/lib6.so python tools/c64_probe.py
# ABL_MACRO=C64XH_ABL selects the bits of the bf16x2 half-K kernel instead.  The product library is never touched.
set -e
cd "$(dirname "$0")/.."
python -c "from wsovod_amd import build; build.build()"
mkdir -p wsovod_amd/lib/abl
objs=$(ls wsovod_amd/csrc/build/*.o | grep -v "/gemm.o")
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -D${ABL_MACRO:-C64P_ABL}=$n -c wsovod_amd/csrc/gemm.hip -o /tmp/gemm_abl$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o wsovod_amd/lib/abl/lib$n.so /tmp/gemm_abl$n.o $objs
done
ls -la wsovod_amd/lib/abl
