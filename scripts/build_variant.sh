#!/bin/bash
# experiment build of the library: scripts/build_variant.sh <name> [-D... | -mllvm ...]   -> lepton_amd/liblepton_<name>.so (git-ignored; travels to
# the GPU box).  Only the two HIP sources are compiled with the extra flags; the host objects are the product build's (run build() first).
set -e
name=$1; shift
cd "$(dirname "$0")/.."
C=lepton_amd/csrc; O=lepton_amd/build/obj; V=lepton_amd/build/variant_$name
mkdir -p $V
# (LEP_SU=1: lep_gpu.hip with -mllvm -structurizecfg-skip-uniform-regions=1 -- measured -1.2 % on the decoder, and the lane-per-unit scan
#  encoder's restart-interval output comes out wrong under it: __graft_entry__.py, profiles/r12d_*; experiments only)
SU=""; [ -n "${LEP_SU:-}" ] && SU="-mllvm -structurizecfg-skip-uniform-regions=1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip $SU "$@" -c $C/lep_gpu.hip -o $V/lep_gpu.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip "$@" -c $C/lep_batch.hip -o $V/lep_batch.o &
wait
third=""
grep -q LEP_HAVE_BROTLI_ENC $O/lep_container.cc.o.flags 2>/dev/null && third="$third $O/brotli/*.o"
if grep -q LEP_PINNED_ZLIB $O/lep_container.cc.o.flags 2>/dev/null; then third="$third $O/zlib/*.o"; else third="$third -lz"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o lepton_amd/liblepton_$name.so $V/lep_gpu.o $V/lep_batch.o $O/lep_api.cc.o $O/jpeg_scan.cc.o \
  $O/jpeg_progressive.cc.o $O/lep_container.cc.o $O/jpeg_recode.cc.o $O/lep_serve.cc.o $third -ldl -lpthread
echo built lepton_amd/liblepton_$name.so
