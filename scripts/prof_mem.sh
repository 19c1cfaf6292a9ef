# usage: bash scripts/prof_mem.sh <outdir-name> ; memory-path counters of the compositing kernels (latency per vector-memory
# instruction = SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM, texture-addresser busy, L1 / L2 hit rates)
set -e
OUT=gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-early-out --no-tet"
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -- $B > /dev/null 2> $OUT/m1.err || true
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc2 -- $B > /dev/null 2> $OUT/m2.err || true
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/pmc3 -- $B > /dev/null 2> $OUT/m3.err || true
python3 scripts/pmc_summary.py $OUT > $OUT/mem_summary.txt
grep "tri_\|sort" $OUT/mem_summary.txt
tail -3 $OUT/m1.err $OUT/m2.err $OUT/m3.err
