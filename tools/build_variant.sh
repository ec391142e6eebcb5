#!/bin/bash
# builds a library variant with extra -D flags for lanes.hip into tools/_variants/<name>.so (experiments; tools/sweep_variants.sh runs them)
#   usage: tools/build_variant.sh <name> -DMSBWT_X=1 ...
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
mkdir -p tools/_variants /tmp/variant_$NAME
OBJS=""
for f in rust-msbwt_amd/build/*.hip.o rust-msbwt_amd/build/*.cpp.o; do
  b=$(basename $f)
  if [ "$b" = lanes.hip.o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c rust-msbwt_amd/csrc/lanes.hip -o /tmp/variant_$NAME/$b
    OBJS="$OBJS /tmp/variant_$NAME/$b"
  else OBJS="$OBJS $f"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_variants/$NAME.so $OBJS -lpthread -ldl
echo built tools/_variants/$NAME.so
