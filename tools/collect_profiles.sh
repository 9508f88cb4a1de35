# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2. PMC passes over the dominant kernel (injected attention, N=4096 d=40 B=1), each its own run
# The raw traces stay on the box (they exceed what a call brings back); tools/summarise_profiles.py turns them
# into the summaries under gpurun_out/${TAG}_summary, which are then copied into profiles/ by hand.
set -e
TAG=${1:-r1}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench --output-format csv -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-block-skip --no-content-cache --no-feature-scale --no-sweep --no-default-queues > $OUT/bench_under_rocprof.json 2> $OUT/bench.err
P=/root/repo/tools/prof_attn.py
timeout -s KILL 90 rocprofv3 --kernel-trace --stats -d $OUT/attn -o attn --output-format csv -- python3 $P 4096 40 1 20 > $OUT/attn.txt 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_ANY -d $OUT/p1 --output-format csv -- python3 $P 4096 40 1 3 > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -d $OUT/p2 --output-format csv -- python3 $P 4096 40 1 3 > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $OUT/p3 --output-format csv -- python3 $P 4096 40 1 3 > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE -d $OUT/p4 --output-format csv -- python3 $P 4096 40 1 3 > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/p5 --output-format csv -- python3 $P 4096 40 1 3 > /dev/null 2>&1
python3 /root/repo/tools/summarise_profiles.py $TAG /root/repo/gpurun_out/${TAG}_summary > $OUT/../${TAG}_summary.log 2>&1
cp $OUT/bench.err /root/repo/gpurun_out/${TAG}_summary/bench.err 2>/dev/null || true
rm -rf $OUT
echo collected
