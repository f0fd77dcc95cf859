#!/bin/bash
# Build a variant of libnbody_hip.so with extra -D flags for nb_tree.hip into
# wgpu_n_body_amd/_variants/<name>.so (git-ignored; travels to the GPU box; use with NB_LIB=...).
#   tools/build_variant.sh w6 -DNB_CELL_STACK=768 -DNB_WALK_MIN_WAVES=6
set -e
cd "$(dirname "$0")/.."
name=$1; shift
P=wgpu_n_body_amd
mkdir -p $P/_variants /tmp/nbv_$name
python -m wgpu_n_body_amd.build > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$P/csrc "$@" -c $P/csrc/nb_tree.hip -o /tmp/nbv_$name/nb_tree.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $P/_variants/$name.so $P/_build/nb_naive.o /tmp/nbv_$name/nb_tree.o $P/_build/nb_abi.o $P/_build/nb_group.o $P/_build/nb_inits.o
echo $P/_variants/$name.so
