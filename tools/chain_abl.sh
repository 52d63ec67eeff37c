#!/bin/bash
# Developer ablation builds of the chained 16-bit head kernel (timing only, results wrong): one probe library per CIAOSR_CHAIN_ABL mask.
#   bash tools/chain_abl.sh 0 64 128 ...   ->  ciaosr_amd/csrc/libciaosr_hip_abl<mask>.so   (needs `make -C ciaosr_amd/csrc probe` first)
# Run:  CIAOSR_HIP_LIB=ciaosr_amd/csrc/libciaosr_hip_abl64.so python tools/chain_probe.py 192 f16
set -e
cd "$(dirname "$0")/../ciaosr_amd/csrc"
mkdir -p build_abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -I../../include -DCIAOSR_PROBE"
OTHERS=$(ls build_probe/*.o | grep -v head_chain_f16)
for m in "$@"; do
    /opt/rocm/bin/hipcc $FLAGS $EXTRA -DCIAOSR_F16=1 -DCIAOSR_CHAIN_ABL=$m -c head_chain_h16.hip -o build_abl/head_chain_f16_$m.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libciaosr_hip_abl$m.so $OTHERS build_abl/head_chain_f16_$m.o
    echo built libciaosr_hip_abl$m.so
done
