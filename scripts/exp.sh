python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --stages 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['grad_max_abs_err'], {k:v for k,v in d['stages_ms'].items() if 'tri' in k})"
DMR_HIPCC_FLAGS="-DDMR_BWD_CHUNK=64" python -m dmesh_renderer_amd.build --force > /dev/null 2>&1
python bench.py --stages --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v for k,v in d['stages_ms'].items() if 'tri' in k})"
