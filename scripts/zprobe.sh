# usage: bash scripts/zprobe.sh <variant> [<variant> ...]   -- the isolated conv timings (zero / post-ReLU-like data) per library variant
cd $GRAFT_REPO_ROOT
PKG=neural_sound_generation_amd
cp $PKG/libnsg.so /tmp/libnsg_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/libnsg_base.so $PKG/libnsg.so; else cp _exp/libnsg_$v.so $PKG/libnsg.so; fi
  echo "== $v"; python scripts/patch_power_probe.py 2>/dev/null | grep -E "back to back" | grep -E "zeros x|relu"
done
cp /tmp/libnsg_base.so $PKG/libnsg.so
