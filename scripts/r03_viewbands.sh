# (view, band) segments: GPU tests of the sharded path, then all eight ranks of C5 cut eight ways + quarters (kernel sums per rank)
set -e
mkdir -p gpurun_out/vb
timeout -k 10 900 python -m pytest tests/test_sharded_gpu.py -x -q > gpurun_out/vb/pytest.log 2>&1 || { tail -30 gpurun_out/vb/pytest.log; exit 1; }
tail -3 gpurun_out/vb/pytest.log
SPECS="0/8 1/8 2/8 3/8 4/8 5/8 6/8 7/8 0/4 1/4 2/4 3/4" bash scripts/prof_shard.sh vb/c5 C5 view_bands
