# Round-5 opener (prepared at the end of round 4, when no GPU time was left): A/B of the activation-prefetch experiment of igemm320.hip
# (-DMOFA_X_PREFETCH=<K tiles ahead>).  One short gpurun call:
#     gpurun --timeout 600 -- 'bash tools/r05_probes.sh'
# Builds happen HERE (no GPU needed) before the call:
#     for n in 2 3 4; do python -m mofa_video_amd._build --variant xpf$n -DMOFA_X_PREFETCH=$n; done
#     python -m mofa_video_amd._build --variant xrpf3 -DMOFA_X_PREFETCH=3 -DMOFA_R_PREFETCH      # + residual-tile lines in the other parity's slot
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
SH="ff out,proj,qkv,conv(3,1,1)"
timeout 200 python tools/igemm_tiles_bench.py --tiles 320p --only "$SH" --rounds 3 > $O/r05_xpf_base.log 2>&1
for n in 2 3 4; do
  [ -f tools/libmofa_hip_xpf$n.so ] || continue
  timeout 200 python tools/igemm_tiles_bench.py --tiles 320p --only "$SH" --rounds 3 --lib tools/libmofa_hip_xpf$n.so > $O/r05_xpf_$n.log 2>&1
done
[ -f tools/libmofa_hip_xrpf3.so ] && timeout 200 python tools/igemm_tiles_bench.py --tiles 320p --only "$SH" --rounds 3 --lib tools/libmofa_hip_xrpf3.so > $O/r05_xpf_r3.log 2>&1
for f in $O/r05_xpf_base.log $O/r05_xpf_[234].log $O/r05_xpf_r3.log; do echo "== $f"; grep -v amdgpu.ids $f | cut -c1-100; done
# parity of the variant on the tile tests (every epilogue kind x mode on the 256x320 tile) before believing any number
for v in xpf3 xrpf3; do [ -f tools/libmofa_hip_$v.so ] && MOFA_HIP_LIB=$GRAFT_REPO_ROOT/tools/libmofa_hip_$v.so timeout 300 python -m pytest tests/test_igemm_tiles_gpu.py tests/test_gn_stats_gpu.py -x -q -k "256x320 or stats" 2>&1 | tail -3; done
