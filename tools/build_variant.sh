#!/bin/bash
# A second build of libicelk.so from the sources as they are NOW (plus -D flags), for A/B runs on one GPU box:
#   tools/build_variant.sh <name> [extra hipcc flags]   ->  variants/libicelk_<name>.so   (select with ICELK_LIBRARY=...)
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
mkdir -p variants/obj_$NAME
SRC="icelk_abi k_image k_pyramid k_lk k_lk_fast k_lk_multi k_corners k_corners_fast k_sort k_tail k_tracks k_utm k_mask k_grid"
for s in $SRC; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-result "$@" \
      -c iceberg_tracking_code_amd/csrc/$s.hip -o variants/obj_$NAME/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libicelk_$NAME.so variants/obj_$NAME/*.o
rm -rf variants/obj_$NAME
ls -la variants/libicelk_$NAME.so
