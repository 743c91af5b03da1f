# build the CURRENT sources with extra compiler flags into ab_live/libltg_<name>.so for same-box A/B runs (LTG_HIP_LIB=...); ab_live/ is git-ignored but
# travels to the GPU box (build_ab/, the round-1-4 archive, is in .gpurunignore): delete what a round no longer needs
# usage: bash scripts/build_variant.sh ieee -DLTG_ADAM_IEEE
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/ab_live
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 "$@" -shared -o $ROOT/ab_live/libltg_$NAME.so $ROOT/long-tail-gan_amd/csrc/ltg_kernels.hip
ls -la $ROOT/ab_live/libltg_$NAME.so
