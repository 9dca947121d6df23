#!/bin/bash
# Builds time-gan_amd/_tgru_torch.so, the C++ autograd binding over libtgru.so (plain g++: the file has no device
# code).  Run from anywhere; needs libtgru.so built first (make -C time-gan_amd/csrc).
set -e
here=$(cd "$(dirname "$0")" && pwd)
pkg=$(dirname "$here")
root=$(dirname "$pkg")
TI=$(python3 -c "import torch,os;print(os.path.dirname(torch.__file__))")
PYI=$(python3 -c "import sysconfig;print(sysconfig.get_paths()['include'])")
PBI=$(python3 -c "import pybind11;print(pybind11.get_include())")
ABI=$(python3 -c "import torch;print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")
g++ -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -DTORCH_EXTENSION_NAME=_tgru_torch \
    -D_GLIBCXX_USE_CXX11_ABI=$ABI -I"$TI/include" -I"$TI/include/torch/csrc/api/include" -I/opt/rocm/include \
    -I"$PYI" -I"$PBI" -I"$root/include" "$here/tgru_torch.cpp" -o "$pkg/_tgru_torch.so" \
    -L"$TI/lib" -lc10 -lc10_hip -ltorch -ltorch_cpu -ltorch_hip -ltorch_python -L"$pkg" -l:libtgru.so \
    -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$TI/lib"
