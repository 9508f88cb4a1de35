# usage: bash tools/tune_plain.sh N Nk d B "ks nw" ...
N=$1; NK=$2; D=$3; B=$4; shift 4
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg
  export SID_TUNE_KS=$1 SID_TUNE_NW=$2
  rm -rf /tmp/tt && timeout -s KILL 90 rocprofv3 --kernel-trace --stats -d /tmp/tt -o t --output-format csv -- python3 /root/repo/tools/prof_plain.py $N $NK $D $B 20 > /dev/null 2>&1 || exit 1
  echo "== plain N=$N Nk=$NK d=$D B=$B KS=$1 NW=$2"
  python3 /root/repo/tools/kstats.py /tmp/tt/t_kernel_stats.csv at::native | grep -v "^kernel"
done
