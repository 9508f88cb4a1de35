# usage: bash tools/ab3.sh name1 name2 ...   (tools/_bin/<name>.so from tools/build_ab.sh; "new" = the in-tree build)
for rep in 1 2; do
for lib in "$@"; do
  if [ $lib = new ]; then unset SID_LIB_OVERRIDE; else export SID_LIB_OVERRIDE=$PWD/tools/_bin/$lib.so; fi
  for a in "4096 40 1 200 f16" "4096 40 1 200 bf16" "1024 80 1 200 f16" "4096 40 1 200 f16 2"; do echo -n "$lib: "; timeout -k 10 60 python3 tools/prof_call.py $a 2>/dev/null || exit 1; done
done
done
