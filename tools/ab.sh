# A/B of two library builds on one box: exp/old.so (build it from an earlier commit: git archive + hipcc, see tools/build_exp.sh) against the in-tree build.
for lib in old new; do
  if [ $lib = old ]; then export SID_LIB_OVERRIDE=$PWD/exp/old.so; else unset SID_LIB_OVERRIDE; fi
  echo -n "$lib 4096/40 B=1: "; timeout -k 10 60 python3 tools/prof_attn.py 4096 40 1 50 2>/dev/null || exit 1
  echo -n "$lib 4096/40 B=2: "; timeout -k 10 60 python3 tools/prof_attn.py 4096 40 2 30 2>/dev/null || exit 1
  echo -n "$lib 4096/40 B=3: "; timeout -k 10 60 python3 tools/prof_attn.py 4096 40 3 30 2>/dev/null || exit 1
  echo -n "$lib 4096/40 B=4: "; timeout -k 10 60 python3 tools/prof_attn.py 4096 40 4 20 2>/dev/null || exit 1
  echo -n "$lib 1024/80: "; timeout -k 10 60 python3 tools/prof_attn.py 1024 80 1 50 2>/dev/null || exit 1
  echo -n "$lib 4096/80: "; timeout -k 10 60 python3 tools/prof_attn.py 4096 80 1 20 2>/dev/null || exit 1
  echo -n "$lib 16384/40: "; timeout -k 10 60 python3 tools/prof_attn.py 16384 40 1 5 2>/dev/null || exit 1
done
if [ -n "$AB_SWEEP" ]; then
  for w in 1 2 3 4 5 6 8; do echo -n "new W=$w: "; SID_TUNE_W=$w timeout -k 10 60 python3 tools/prof_attn.py 4096 40 1 50 2>/dev/null || exit 1; done
  for w in 1 2 3 4 6; do echo -n "new 1024/80 W=$w: "; SID_TUNE_W=$w timeout -k 10 60 python3 tools/prof_attn.py 1024 80 1 50 2>/dev/null || exit 1; done
fi
