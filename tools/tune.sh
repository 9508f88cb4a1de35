for cfg in "1 2" "1 4" "2 2" "2 4" "2 1" "3 4" "4 4"; do set -- $cfg; echo "KS=$1 NW=$2"; SID_TUNE_KS=$1 SID_TUNE_NW=$2 timeout -k 10 60 python3 tools/prof_attn.py 4096 40 1 20 || exit 1; done
for B in 2 4; do for cfg in "1 4" "1 2" "2 4"; do set -- $cfg; echo "B=$B KS=$1 NW=$2"; SID_TUNE_KS=$1 SID_TUNE_NW=$2 timeout -k 10 60 python3 tools/prof_attn.py 4096 40 $B 20 || exit 1; done; done
