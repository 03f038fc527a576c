# Build (here, cross-compiling; the binaries travel with the gpurun snapshot) the tools of the experimental 3-stage igemm
# kernel (mofa_video_amd/csrc/igemm_ring3.inc).   usage: bash tools/build_ring3_tools.sh
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -DMOFA_IGEMM_RING3 -I include"
/opt/rocm/bin/hipcc $F tools/igemm_ring3_check.hip -o tools/igemm_ring3_check.bin &
/opt/rocm/bin/hipcc $F -DMOFA_IGEMM_TRACE tools/igemm_trace.hip -o tools/igemm_trace_ring3.bin &
wait
ls -la tools/igemm_ring3_check.bin tools/igemm_trace_ring3.bin
