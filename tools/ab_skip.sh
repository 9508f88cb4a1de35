# default path before/after the block-skipping code went in, then the flag itself on feature-scale data
for rep in 1 2; do
for lib in head new; do
  if [ $lib = new ]; then unset SID_LIB_OVERRIDE; else export SID_LIB_OVERRIDE=$PWD/ab/$lib.so; fi
  for a in "4096 40 1 200" "1024 80 1 200"; do echo -n "$lib: "; timeout -k 10 60 python3 tools/prof_call.py $a 2>/dev/null || exit 1; done
done
done
unset SID_LIB_OVERRIDE
for m in "3 disc" "2 disc" "1 disc" "3 halves"; do
for a in "4096 40 1 200 bf16 0" "4096 40 1 200 bf16 1" "4096 40 16 20 bf16 0" "4096 40 16 20 bf16 1"; do echo -n "[$a $m] "; timeout -k 10 60 python3 tools/prof_call.py $a $m 2>/dev/null || exit 1; done
done
