# Same-device A/B of two library builds (devices of the pool differ by +-5 %, more than most changes are worth):
#   bash scripts/ab.sh build [rev]     builds ab_build/old (rev, default HEAD) and ab_build/new (working tree); here, no GPU
#   bash scripts/ab.sh run <script + args>   on the GPU box: old, new, old, new, ... three times each
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=voxel-based-3d-reconstruction_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -I/opt/rocm/include"
if [ "$1" = build ]; then
  REV=${2:-HEAD}
  rm -rf $ROOT/ab_build; mkdir -p $ROOT/ab_build/old $ROOT/ab_build/new
  # the old revision's sources go to a scratch directory and are deleted again: only the two libraries stay in the tree
  TMP=$(mktemp -d); mkdir -p $TMP/pkg/csrc $TMP/include
  git -C $ROOT show $REV:include/voxcarve.h > $TMP/include/voxcarve.h                    # (voxcarve.hip includes ../../include/voxcarve.h)
  for f in $(git -C $ROOT ls-tree --name-only $REV $SRC/); do git -C $ROOT show $REV:$f > $TMP/pkg/csrc/$(basename $f); done
  /opt/rocm/bin/hipcc $FLAGS -o $ROOT/ab_build/old/libvoxcarve.so $TMP/pkg/csrc/voxcarve.hip -ldl
  rm -rf $TMP
  /opt/rocm/bin/hipcc $FLAGS -o $ROOT/ab_build/new/libvoxcarve.so $ROOT/$SRC/voxcarve.hip -ldl
  echo built
else
  shift
  for i in 1 2 3; do
    for v in old new; do
      echo -n "$v: "; VOXCARVE_LIB=$ROOT/ab_build/$v/libvoxcarve.so timeout -k 10 300 python "$@" | tail -1
    done
  done
fi
