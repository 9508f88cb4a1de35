for nw in 2 4 8; do for w in 1 2 3 6; do echo -n "NW=$nw W=$w: "; SID_TUNE_NW=$nw SID_TUNE_W=$w timeout -k 10 60 python3 tools/prof_call.py 1024 80 1 200 bf16 2 2>/dev/null || exit 1; done; done
echo -n "default: "; timeout -k 10 60 python3 tools/prof_call.py 1024 80 1 200 bf16 2
echo -n "default no-throughput: "; timeout -k 10 60 python3 tools/prof_call.py 1024 80 1 200 bf16 0
