#!/bin/bash
# tools/build_variants.sh NAME "-DFLAG=.. -DFLAG2=.." -- A/B build of the graph kinds with extra macros:
# writes variants/libfundsp_hip_NAME.so (select it with FUNDSP_HIP_LIB=...).  Design tool, not part of the product.
set -e
cd "$(dirname "$0")/../fundsp_amd/csrc"
NAME=$1; shift
mkdir -p ../../variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-unused-value"
/opt/rocm/bin/hipcc $FLAGS $@ -c fd_kinds_graph.hip -o /tmp/fd_kinds_graph_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libfundsp_hip_$NAME.so fd_capi.o fd_kinds_leaf.o /tmp/fd_kinds_graph_$NAME.o fd_fdn.o fd_jit.o fd_comm.o fd_rust.o -lhiprtc -lrccl -ldl
echo built variants/libfundsp_hip_$NAME.so
