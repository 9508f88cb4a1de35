# Block skipping with the content block cached: the call with and without the flag on one box.
#   bash tools/ab_skip_cached.sh        (prints "cached content block: call ..." lines; flag 0 against flag 1)
set -o pipefail
for cfg in "1 3.0 disc" "1 2.0 disc" "1 1.0 disc" "16 3.0 disc" "16 2.0 disc" "16 1.0 disc" "1 2.0 real:content_car_mask" "16 2.0 real:content_car_mask"; do
  set -- $cfg
  for fl in 0 1; do
    echo -n "B=$1 sigma=$2 $3 flags=$fl: "
    timeout -k 10 120 python3 tools/prof_call.py 4096 40 $1 $([ $1 = 1 ] && echo 200 || echo 20) f16 $fl $2 $3 2>/dev/null | tr '\n' ' ' || exit 1
    echo
  done
done
