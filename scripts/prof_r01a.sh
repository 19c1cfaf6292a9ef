set -e
mkdir -p gpurun_out/prof_r01a
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01a/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_r01a/bench_trace.json 2> gpurun_out/prof_r01a/trace.err || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/prof_r01a/pmc1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r01a/bench_pmc1.json 2> gpurun_out/prof_r01a/pmc1.err || true
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d gpurun_out/prof_r01a/pmc2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r01a/bench_pmc2.json 2> gpurun_out/prof_r01a/pmc2.err || true
find gpurun_out/prof_r01a -name "*.csv" | head -20
