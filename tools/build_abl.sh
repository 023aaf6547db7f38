#!/bin/bash
# tools/build_abl.sh NAME "-DFLAG ..." : builds an A/B variant of the engine library to tools/_abl/NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
d=tools/_abl/obj_$name; mkdir -p $d
for f in gemm gemm_st gemm_rowln attention elementwise head skinny cls_tail patches metrics mfma_stream engine; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $@ -c vtamiq_amd/csrc/$f.hip -o $d/$f.o &
done
wait; for f in gemm gemm_st gemm_rowln attention elementwise head skinny cls_tail patches metrics mfma_stream engine; do test -f $d/$f.o || { echo "compile of $f failed"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_abl/$name.so $d/*.o
rm -rf $d
echo built tools/_abl/$name.so
