# usage: bash scripts/prof_pmc.sh <outdir-name> ; per-kernel averages of two --pmc passes -> gpurun_out/<name>/pmc_summary.txt
set -e
OUT=gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-early-out --no-tet > $OUT/bench_pmc1.json 2> $OUT/pmc1.err || true
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-early-out --no-tet > $OUT/bench_pmc2.json 2> $OUT/pmc2.err || true
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt 
