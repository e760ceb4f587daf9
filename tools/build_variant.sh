#!/bin/bash
# tools/build_variant.sh <git-rev|WORK> <name>  -> $AB_OUT/<name>.so, AB_OUT=/tmp/ab_libs unless set  (for A/B of two builds)
# The output lies OUTSIDE the repository on purpose: everything inside it travels to the GPU box with every gpurun call.  Copy the builds a
# measurement needs into ab_ship/ (git-ignored) for that call, name them there (tools/uat_ab.py ab_ship/a.so ab_ship/b.so), delete them afterwards.
set -e
rev=$1; name=$2; root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
if [ "$rev" = "WORK" ]; then cp -r $root/libadsb_amd/csrc $tmp/csrc; cp -r $root/include $tmp/include
else mkdir -p $tmp/csrc $tmp/include; (cd $root && git archive $rev libadsb_amd/csrc include | tar -x -C $tmp); mv $tmp/libadsb_amd/csrc/* $tmp/csrc/; fi
srcs="$tmp/csrc/scan1090.hip $tmp/csrc/capi.cpp $tmp/csrc/resolver1090.cpp"
for f in scan2400.hip transport.cpp adsb1090_gpu_handler.cpp uat978.hip uat978_host.cpp uat978_gpu_handler.cpp; do [ -f $tmp/csrc/$f ] && srcs="$srcs $tmp/csrc/$f"; done
objs=""
for f in $srcs; do
  extra=""; case $(basename $f) in scan1090.hip|scan2400.hip) extra="-mllvm -amdgpu-atomic-optimizer-strategy=None";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -march=x86-64-v3 -ffp-contract=off $extra $EXTRA_FLAGS -I$tmp/include -I$tmp/csrc -c $f -o $tmp/$(basename $f).o &
  objs="$objs $tmp/$(basename $f).o"
done
wait
out=${AB_OUT:-/tmp/ab_libs}; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $objs
rm -rf $tmp; echo built $out/$name.so
