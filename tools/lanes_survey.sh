# imgs/s of the hot path by images in flight and forced runs per head (SID_TUNE_W)
# usage (GPU box): bash tools/lanes_survey.sh > gpurun_out/lanes_survey.txt
B="timeout -k 10 200 python3 bench.py --no-e2e --no-cpu-baseline --no-block-skip --no-content-cache --steps 24 --warmup 6"
val() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.1f imgs/s' % d['value'])"; }
for lanes in 3 6; do echo "default lanes=$lanes: $($B --in-flight $lanes | val)" || exit 1; done
for lanes in 3 4 6 8; do echo "SID_TUNE_W=1 lanes=$lanes: $(SID_TUNE_W=1 $B --in-flight $lanes | val)" || exit 1; done
