#!/bin/bash
# experiment build of the library: scripts/build_variant.sh <name> [-D...]   -> lepton_amd/liblepton_<name>.so (git-ignored; travels to the GPU box)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
C=lepton_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o lepton_amd/liblepton_$name.so $C/lep_gpu.hip $C/lep_batch.hip $C/lep_api.cc $C/jpeg_scan.cc \
  $C/jpeg_progressive.cc $C/lep_container.cc $C/jpeg_recode.cc $C/lep_serve.cc -lz -ldl -lpthread
echo built lepton_amd/liblepton_$name.so
