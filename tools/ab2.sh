# A/B of two library builds on one box: tools/_bin/old.so (built from an earlier commit) against the in-tree build.
for rep in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export SID_LIB_OVERRIDE=$PWD/tools/_bin/old.so; else unset SID_LIB_OVERRIDE; fi
  for a in "4096 40 1 200" "1024 80 1 200"; do echo -n "$lib: "; timeout -k 10 60 python3 tools/prof_attn.py $a 2>/dev/null || exit 1; done
done
done
