#!/usr/bin/env bash
# Build libadaptigraph_hip.so for gfx950 (MI355X), in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function ${AG_EXTRA_FLAGS:-}"
mkdir -p build
# per-file extras: the fused MLP chains are scheduled for ILP (hipcc's default strategy leaves ~0.9 % on k_edge_enc and
# ~0.5 % on k_node_prop: A/B on the same box); scheduling only, results are bit-identical
declare -A PERFILE=( [ag_mlp]="-mllvm -amdgpu-sched-strategy=max-ilp" )
for f in ag_edges ag_rules ag_mlp ag_lat ag_graph ag_cost ag_mppi ag_api; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ ag_common.h -nt build/$f.o ] || [ ../../include/adaptigraph_amd.h -nt build/$f.o ] || [ build.sh -nt build/$f.o ]; then
    $HIPCC $FLAGS ${PERFILE[$f]:-} -c $f.hip -o build/$f.o
  fi
done
$HIPCC -shared -fPIC --offload-arch=gfx950 build/ag_edges.o build/ag_rules.o build/ag_mlp.o build/ag_lat.o build/ag_graph.o build/ag_cost.o build/ag_mppi.o build/ag_api.o -o libadaptigraph_hip.so
echo "built $(pwd)/libadaptigraph_hip.so"
