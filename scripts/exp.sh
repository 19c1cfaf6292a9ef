for f in "-DDMR_FWD_WAVES=5" "-DDMR_FWD_WAVES=6" "-DDMR_FWD_WAVES=5 -DDMR_PIX_WAVES=5"; do
DMR_HIPCC_FLAGS="$f" python -m dmesh_renderer_amd.build --force > /dev/null 2>&1
echo "$f"; python bench.py --stages --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v for k,v in d['stages_ms'].items() if 'tri' in k})"
done
