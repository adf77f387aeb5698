#!/bin/bash
# usage: tools/abbuild.sh NAME [-DFLAG ...]   ->  tools/ab/libvp_hip_NAME.so (a side-by-side build of libvp_hip.so for tools/abbench.py; tools only)
# Objects are cached per source file and flag set under tools/ab/obj, so a variant that touches one file recompiles one file.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
flags="$*"
tag=$(echo "$flags" | md5sum | cut -c1-8)
mkdir -p tools/ab/obj
objs=()
for f in vision-processor_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  [ "$b" = vp_multi ] && continue
  h=$(cat $f vision-processor_amd/csrc/*.h include/vp_hip.h | md5sum | cut -c1-12)
  o=tools/ab/obj/$b.$tag.$h.o
  if [ ! -f $o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -Wall -Wno-unused-function -w $flags -c $f -o $o &
  fi
  objs+=($o)
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/ab/libvp_hip_$name.so "${objs[@]}"
echo "tools/ab/libvp_hip_$name.so"
