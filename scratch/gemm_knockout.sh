#!/bin/bash
# Builds diagnostic copies of the library whose GEMM kernel lacks one part at a time (TGRU_GEMM_KO bit mask,
# csrc/gemm.hip) into scratch/ko/, for scratch/gemm_knockout.py to time on the GPU.
set -e
cd "$(dirname "$0")/../time-gan_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../scratch/ko
for ko in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DTGRU_GEMM_KO=$ko $TGRU_KO_EXTRA -c gemm.hip -o /tmp/gemm_ko$ko.o
  objs=$(ls *.o | grep -v '^gemm.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/ko/libtgru_ko$ko.so /tmp/gemm_ko$ko.o $objs
done
