# Anatomy of lin320 by removal: timing-only builds (wrong results; the data flow stays alive), one -D switch each.
#   bash tools/lin320_anatomy.sh build   (build container: cross-compiles tools/libmofa_hip_lin_*.so)
#   bash tools/lin320_anatomy.sh run     (GPU box)
V="nostore:LIN_T_NOSTORE noepi:LIN_T_NOEPI nobar:LIN_T_NOBAR nodma:LIN_T_NODMA noepibar:LIN_T_NOEPI+LIN_T_NOBAR nomem:LIN_T_NOEPI+LIN_T_NOBAR+LIN_T_NODMA"
if [ "$1" = build ]; then
  for v in $V; do n=${v%%:*}; d=$(echo ${v#*:} | sed 's/+/ -D/g'); (python -m mofa_video_amd._build --incremental --variant lin_$n -D$d 2>&1 | tail -1 | cut -c1-80) & done; wait
else
  python tools/lin320_bench.py --frames 50 2>&1 | grep "N = 960\|normed" | sed 's/^/shipped  /'
  for v in $V; do n=${v%%:*}; MOFA_HIP_LIB=tools/libmofa_hip_lin_$n.so python tools/lin320_bench.py --frames 50 2>&1 | grep "N = 960\|normed" | sed "s/^/$n /"; done
fi
