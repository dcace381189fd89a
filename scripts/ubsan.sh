# Host-side UndefinedBehaviorSanitizer run of the GPU suite (GPU ASan / xnack are not available on this pool).
# Builds a second libvoxcarve.so with -Xarch_host -fsanitize=undefined under ubsan_build/ (git-ignored, travels with gpurun)
# and points the binding at it (VOXCARVE_LIB); run the printed command on the GPU box.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/ubsan_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -I/opt/rocm/include \
    -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined \
    -o $ROOT/ubsan_build/libvoxcarve.so $ROOT/voxel-based-3d-reconstruction_amd/csrc/voxcarve.hip -ldl
echo 'RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so); LD_PRELOAD=$RT VOXCARVE_LIB=$PWD/ubsan_build/libvoxcarve.so UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python -m pytest tests -m gpu -x -q -k "not 1024 and not u32_limit and not bench and not config5_full"'
