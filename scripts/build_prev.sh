# build the library of a commit (default HEAD) into build_ab/libltg_prev.so for same-box A/B runs (scripts/ab.sh)
set -e
REV=${1:-HEAD}
ROOT=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
mkdir -p $T/long-tail-gan_amd/csrc $T/include $ROOT/build_ab
for f in $(git -C $ROOT ls-tree --name-only $REV long-tail-gan_amd/csrc/ include/); do git -C $ROOT show $REV:$f > $T/$f; done
make -C $T/long-tail-gan_amd/csrc >/dev/null
cp $T/long-tail-gan_amd/libltg_hip.so $ROOT/build_ab/libltg_prev.so
rm -rf $T
ls -la $ROOT/build_ab/libltg_prev.so
