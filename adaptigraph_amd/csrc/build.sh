#!/usr/bin/env bash
# Build libadaptigraph_hip.so for gfx950 (MI355X), in-tree.  hipcc cross-compiles without a GPU.
#   build.sh         the product library
#   build.sh diag    also libadaptigraph_hip_diag.so: the same sources with -DAG_DIAG (in-kernel clock / phase probes and the
#                    injected-failure hook, ag_diag.hip) - used by tools/ and one test, never loaded by the product package
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function ${AG_EXTRA_FLAGS:-}"
# per-file extras: the fused MLP chains are scheduled for ILP (hipcc's default strategy leaves ~0.9 % on k_edge_enc and
# ~0.5 % on k_node_prop: A/B on the same box); scheduling only, results are bit-identical
declare -A PERFILE=( [ag_mlp]="-mllvm -amdgpu-sched-strategy=max-ilp" )
SRCS="ag_edges ag_rules ag_mlp ag_lat ag_graph ag_cost ag_mppi ag_api"
build_variant() {   # $1 = object dir, $2 = extra flags, $3 = output, $4 = extra sources
  mkdir -p "$1"
  local objs=""
  for f in $SRCS $4; do
    if [ ! -f $1/$f.o ] || [ $f.hip -nt $1/$f.o ] || [ ag_common.h -nt $1/$f.o ] || [ ../../include/adaptigraph_amd.h -nt $1/$f.o ] || [ build.sh -nt $1/$f.o ]; then
      $HIPCC $FLAGS $2 ${PERFILE[$f]:-} -c $f.hip -o $1/$f.o
    fi
    objs="$objs $1/$f.o"
  done
  $HIPCC -shared -fPIC --offload-arch=gfx950 $objs -o $3
  echo "built $(pwd)/$3"
}
build_variant build "" libadaptigraph_hip.so ""
if [ "${1:-}" = "diag" ]; then
  build_variant build/diag "-DAG_DIAG" libadaptigraph_hip_diag.so "ag_diag"
fi
