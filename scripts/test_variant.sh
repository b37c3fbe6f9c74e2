# bash scripts/test_variant.sh <variant> <pytest args...>: run tests against _exp/libnsg_<variant>.so
cd $GRAFT_REPO_ROOT
PKG=neural_sound_generation_amd
V=$1; shift
cp $PKG/libnsg.so /tmp/libnsg_keep.so
cp _exp/libnsg_$V.so $PKG/libnsg.so
python -m pytest "$@" 2>&1 | tail -15
cp /tmp/libnsg_keep.so $PKG/libnsg.so
