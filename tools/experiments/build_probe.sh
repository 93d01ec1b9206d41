#!/bin/bash
# diagnostic builds of the library: tools/experiments/build_probe.sh <name> <source.hip> <-DDEFINE[=v]> -> tools/experiments/librpe_probe<name>.so
cd "$(dirname "$0")/../.."
name=$1; src=$2; def=$3
mkdir -p /tmp/probe$name
for f in rpeflow_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ "$b.hip" = "$src" ]; then
    extra=""; [ "$b" = "knn" ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fvisibility=hidden $extra $def -I include -I rpeflow_amd/csrc -c $f -o /tmp/probe$name/$b.o
  else
    cp rpeflow_amd/csrc/build/$b.o /tmp/probe$name/$b.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/librpe_probe$name.so /tmp/probe$name/*.o
ls -la tools/experiments/librpe_probe$name.so
