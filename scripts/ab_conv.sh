cd $GRAFT_REPO_ROOT
PKG=neural_sound_generation_amd
cp $PKG/libnsg.so /tmp/libnsg_base.so
for r in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/libnsg_base.so $PKG/libnsg.so; else cp _exp/libnsg_$v.so $PKG/libnsg.so; fi
  echo "$v: $(python scripts/ab_conv.py 2>/dev/null)"
done; done
cp /tmp/libnsg_base.so $PKG/libnsg.so
