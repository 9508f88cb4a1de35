# imgs/s of the hot path by images in flight and hardware queues per process
# usage (GPU box): bash tools/lanes_survey.sh > gpurun_out/lanes_survey.txt
B="timeout -k 10 200 python3 bench.py --no-e2e --no-cpu-baseline --no-block-skip --no-content-cache --steps 24 --warmup 6"
val() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.1f imgs/s' % d['value'])"; }
for q in 4 8; do for lanes in 3 5 6 7 8; do
  echo "GPU_MAX_HW_QUEUES=$q lanes=$lanes: $($B --hw-queues $q --in-flight $lanes | val)" || exit 1
done; done
