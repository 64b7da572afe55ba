#!/usr/bin/env bash
# Build an EXPERIMENT variant of the library next to the product one: same sources, extra -D flags, own object dir and name.
#   tools/build_experiment.sh <name> <flags...>      ->  adaptigraph_amd/csrc/libadaptigraph_hip_<name>.so
# Loaded with ADAPTIGRAPH_AMD_LIB=<path> by tools/ only; never by the product package or the tests.
set -euo pipefail
name=$1; shift
cd "$(dirname "$0")/../adaptigraph_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $*"
mkdir -p build/$name
objs=""
for f in ag_edges ag_rules ag_mlp ag_lat ag_graph ag_cost ag_mppi ag_api; do
  extra=""; [ $f = ag_mlp ] && extra="-mllvm -amdgpu-sched-strategy=max-ilp"
  $HIPCC $FLAGS $extra -c $f.hip -o build/$name/$f.o &
  objs="$objs build/$name/$f.o"
done
wait
$HIPCC -shared -fPIC --offload-arch=gfx950 $objs -o libadaptigraph_hip_$name.so
echo "built $(pwd)/libadaptigraph_hip_$name.so"
