# PMC passes over the injected attention at one shape; each pass is its own run.
# usage: bash tools/pmc_attn.sh N d B outdir
set -e
N=$1; D=$2; B=$3; OUT=$4
cd /tmp && export TMPDIR=/tmp
P=/root/repo/tools/prof_attn.py
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_ANY -d $OUT/p1 --output-format csv -- python3 $P $N $D $B 3 > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -d $OUT/p2 --output-format csv -- python3 $P $N $D $B 3 > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE -d $OUT/p3 --output-format csv -- python3 $P $N $D $B 3 > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT/p4 --output-format csv -- python3 $P $N $D $B 3 > /dev/null 2>&1
cd /root/repo && python3 tools/pmc_summary.py $OUT flash_kernel
