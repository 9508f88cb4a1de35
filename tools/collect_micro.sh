# Microbenchmarks and diagnostic-build runs that back the analysis of the flash kernel (gpurun, repo root):
#   issue-rate microbenchmarks (tools/ubench_*.hip, built into exp/) and per-workgroup clock stamps
#   (exp/diag.so from tools/build_diag.sh).  Output: gpurun_out/$TAG/micro_*.txt
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 120 exp/ubench_issue2 > $OUT/micro_issue.txt 2>&1 || exit 1
timeout -k 10 120 exp/ubench_slot > $OUT/micro_slot.txt 2>&1 || exit 1
: > $OUT/micro_diag.txt
for cfg in "4096 40 1" "4096 40 4" "16384 40 1" "1024 80 1"; do
  echo "== diag_flash $cfg" >> $OUT/micro_diag.txt
  timeout -k 10 100 python3 tools/diag_flash.py $cfg >> $OUT/micro_diag.txt 2>/dev/null || exit 1
done
echo micro collected
