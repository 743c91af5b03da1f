# build the CURRENT sources with extra compiler flags into build_ab/libltg_<name>.so for same-box A/B runs (LTG_HIP_LIB=...)
# usage: bash scripts/build_variant.sh ieee -DLTG_ADAM_IEEE
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/build_ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -shared -o $ROOT/build_ab/libltg_$NAME.so $ROOT/long-tail-gan_amd/csrc/ltg_kernels.hip
ls -la $ROOT/build_ab/libltg_$NAME.so
