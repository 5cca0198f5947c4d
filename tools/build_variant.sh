#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> "<extra -D flags>"
# Builds build/variants/lib<name>.so: <file.hip> recompiled with the extra flags, every other object from the normal
# build.  Select it at run time with WMF_HIP_LIB=build/variants/lib<name>.so (kernel tuning experiments only).
set -e
name=$1; src=$2; extra=$3
root=$(cd "$(dirname "$0")/.." && pwd)
c=$root/recmodel_amd/csrc
mkdir -p $root/build/variants
make -s -C $c
[ "$src" = "wmf_rowsplit.hip" ] && [ -z "$RS_KEEP_SLP" ] && extra="$extra -fno-slp-vectorize"      # as the Makefile does (RS_KEEP_SLP=1: lab)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wall -Wno-unused-function $extra -c $c/$src -o $root/build/variants/$name.o
if [ "$src" = "wmf_directl.hip" ]; then python3 $root/tools/check_inflight_regs.py $c/$src $extra; fi
objs=""
for o in $c/*.o; do
  [ "$(basename $o)" = "${src%.hip}.o" ] && continue
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/variants/lib$name.so $objs $root/build/variants/$name.o -Wl,-rpath,/opt/rocm/lib
echo built build/variants/lib$name.so
