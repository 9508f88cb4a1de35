for cfg in "4 0" "8 0" "8 2" "8 3" "8 4" "8 6"; do set -- $cfg
  export SID_TUNE_NW=$1; if [ $2 = 0 ]; then unset SID_TUNE_W; else export SID_TUNE_W=$2; fi
  echo -n "NW=$1 W=$2: "; timeout -k 10 60 python3 tools/prof_attn.py 4096 40 1 50 2>/dev/null | sed 's/.*flash/flash/' || exit 1
done
unset SID_TUNE_W
for nw in 4 8; do export SID_TUNE_NW=$nw
echo -n "NW=$nw B=4: "; timeout -k 10 60 python3 tools/prof_attn.py 4096 40 4 20 2>/dev/null | sed 's/.*flash/flash/'
echo -n "NW=$nw 16384: "; timeout -k 10 60 python3 tools/prof_attn.py 16384 40 1 5 2>/dev/null | sed 's/.*flash/flash/'
echo -n "NW=$nw 4096/80: "; timeout -k 10 60 python3 tools/prof_attn.py 4096 80 1 20 2>/dev/null | sed 's/.*flash/flash/'
done
