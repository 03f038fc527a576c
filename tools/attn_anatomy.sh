# Anatomy of the spatial attention kernel by removal (round-5 verdict item 3).  Timing-only builds of libmofa_hip.so (wrong results, the
# data flow from Q.K^T through the softmax to P.V stays intact so that no MFMA is dead code), one -D switch each; run on the GPU box:
#   bash tools/attn_anatomy.sh build   (in the build container: cross-compiles the variants into tools/libmofa_hip_att_*.so)
#   bash tools/attn_anatomy.sh run     (on the GPU box)
V="noreadk:ATT_T_NOREADK noreadv:ATT_T_NOREADV nodma:ATT_T_NODMA nobar:ATT_T_NOBAR noexp:ATT_T_NOEXP nosum:ATT_T_NOSUM"
V="$V noreads:ATT_T_NOREADK+ATT_T_NOREADV nomem:ATT_T_NOREADK+ATT_T_NOREADV+ATT_T_NODMA+ATT_T_NOBAR novalu:ATT_T_NOEXP+ATT_T_NOSUM"
if [ "$1" = build ]; then
  for v in $V; do n=${v%%:*}; d=$(echo ${v#*:} | sed 's/+/ -D/g'); (python -m mofa_video_amd._build --incremental --variant att_$n -D$d 2>&1 | tail -1) & done; wait
else
  python tools/attn_bench.py --iters 5 2>&1 | grep -E "L0 |L1 " | sed 's/^/shipped   /'
  for v in $V; do n=${v%%:*}; python tools/attn_bench.py --iters 5 --lib tools/libmofa_hip_att_$n.so 2>&1 | grep -E "L0 |L1 " | sed "s/^/$n /"; done
fi
