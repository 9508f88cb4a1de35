# usage: bash tools/tune_trace.sh N d B "ks nw" "ks nw" ...   -> per-kernel average durations from rocprofv3 for each setting
N=$1; D=$2; B=$3; shift 3
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg
  export SID_TUNE_KS=$1 SID_TUNE_NW=$2
  rm -rf /tmp/tt && timeout -s KILL 90 rocprofv3 --kernel-trace --stats -d /tmp/tt -o t --output-format csv -- python3 /root/repo/tools/prof_attn.py $N $D $B 20 > /dev/null 2>&1 || exit 1
  echo "== N=$N d=$D B=$B KS=$1 NW=$2"
  python3 /root/repo/tools/kstats.py /tmp/tt/t_kernel_stats.csv at::native | grep -v "^kernel" 
done
