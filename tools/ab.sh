for lib in old new; do
  if [ $lib = old ]; then export SID_LIB_OVERRIDE=$PWD/exp/old.so; else unset SID_LIB_OVERRIDE; fi
  for cfg in "0 0" "1 4" "2 4" "4 4" "2 2" "4 2"; do set -- $cfg
    if [ $1 = 0 ]; then unset SID_TUNE_KS SID_TUNE_NW; else export SID_TUNE_KS=$1 SID_TUNE_NW=$2; fi
    echo -n "$lib KS=$1 NW=$2: "; timeout -k 10 60 python3 tools/prof_attn.py 4096 40 1 50 2>/dev/null || exit 1
  done
  unset SID_TUNE_KS SID_TUNE_NW
  echo -n "$lib B=4: "; timeout -k 10 60 python3 tools/prof_attn.py 4096 40 4 20 2>/dev/null || exit 1
  echo -n "$lib 1024/80: "; timeout -k 10 60 python3 tools/prof_attn.py 1024 80 1 50 2>/dev/null || exit 1
  echo -n "$lib 4096/80: "; timeout -k 10 60 python3 tools/prof_attn.py 4096 80 1 20 2>/dev/null || exit 1
done
