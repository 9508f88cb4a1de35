# Forced decompositions of the B = 1 injected call at (4096,40): runs per head (SID_TUNE_W) x waves per workgroup
# (SID_TUNE_NW), graph-timed call and flash stage.   usage (GPU box): bash tools/tune_w.sh [compute] > gpurun_out/tune_w.txt
C=${1:-f16}
for rep in 1 2; do
for cfg in "- -" "2 4" "4 4" "8 4" "3 8" "6 8" "2 8" "5 8"; do
  set -- $cfg
  if [ "$1" = "-" ]; then unset SID_TUNE_W SID_TUNE_NW; else export SID_TUNE_W=$1 SID_TUNE_NW=$2; fi
  echo -n "[W=$1 NW=$2 $C] "
  timeout -k 10 60 python3 tools/prof_call.py 4096 40 1 200 $C 0 1 disc 2>/dev/null || exit 1
done
done
