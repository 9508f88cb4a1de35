for lib in old new old new; do
  if [ $lib = old ]; then export SID_LIB_OVERRIDE=$PWD/ab/old.so; else unset SID_LIB_OVERRIDE; fi
  echo "== $lib"; timeout -k 10 60 python3 tools/prof_hf.py 2>/dev/null || exit 1
done
