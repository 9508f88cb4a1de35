for m in "3 disc" "3 real:content_flower_mask" "2 disc" "2 real:content_flower_mask" "1 disc"; do
for a in "4096 40 1 200 bf16 0" "4096 40 1 200 bf16 1" "4096 40 16 20 bf16 0" "4096 40 16 20 bf16 1"; do echo -n "[$a $m] "; timeout -k 10 60 python3 tools/prof_call.py $a $m 2>/dev/null || exit 1; done
done
for a in "4096 40 1 200 f16 0 3 disc" "4096 40 1 200 f16 1 3 disc" "4096 80 1 100 bf16 0 3 disc" "4096 80 1 100 bf16 1 3 disc" "16384 40 1 20 bf16 0 3 disc" "16384 40 1 20 bf16 1 3 disc"; do echo -n "[$a] "; timeout -k 10 60 python3 tools/prof_call.py $a 2>/dev/null || exit 1; done
