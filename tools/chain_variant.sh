#!/bin/bash
# Developer helper: build libciaosr_hip_abl<NAME>.so = the normal library with head_chain_h16.hip replaced by SOURCE (both element types).
#   bash tools/chain_variant.sh NAME SOURCE.hip [extra hipcc flags]      (needs `make -C ciaosr_amd/csrc` first; A/B with tools/chain_ab.py)
set -e
NAME=$1; SRC=$(realpath "$2"); shift 2
cd "$(dirname "$0")/../ciaosr_amd/csrc"
mkdir -p build_abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -I. -I../../include $*"
/opt/rocm/bin/hipcc $FLAGS -DCIAOSR_F16=1 -c "$SRC" -o build_abl/v_${NAME}_f16.o
/opt/rocm/bin/hipcc $FLAGS -DCIAOSR_F16=0 -c "$SRC" -o build_abl/v_${NAME}_bf16.o
OTHERS=$(ls build/*.o | grep -v head_chain_)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libciaosr_hip_abl${NAME}.so $OTHERS build_abl/v_${NAME}_f16.o build_abl/v_${NAME}_bf16.o
echo built libciaosr_hip_abl${NAME}.so
