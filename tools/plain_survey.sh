# stage timings of the single-block call at the workload's shapes, default and forced decompositions
# usage (GPU box): bash tools/plain_survey.sh > gpurun_out/plain_survey.txt
set -e
P="timeout -k 10 120 python3 tools/prof_plain_call.py"
for shape in "4096 40" "1024 80" "256 160" "64 160"; do
  $P $shape 1 200
done
for w in 1 2 3 4; do for nw in 4 8; do
  echo "W=$w NW=$nw"; SID_TUNE_W=$w SID_TUNE_NW=$nw $P 4096 40 1 200
done; done
for w in 1 2 4; do for nw in 2 4; do
  echo "W=$w NW=$nw"; SID_TUNE_W=$w SID_TUNE_NW=$nw $P 1024 80 1 200
done; done
