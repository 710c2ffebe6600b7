#!/bin/bash
# Timing-only experiment builds of the library: tools/build_exp.sh NAME "-DFLAG1 -DFLAG2=.."
set -e
cd "$(dirname "$0")/../tensornetworkforml_amd/csrc"
mkdir -p ../../build_exp
for f in tnml_api kernels_wide kernels_narrow kernels_big; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/opt/rocm/include $2 -c $f.hip -o ../../build_exp/$1_$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_exp/libtnml_$1.so ../../build_exp/$1_*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -f ../../build_exp/$1_*.o
