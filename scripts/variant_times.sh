# usage: bash scripts/variant_times.sh <outname> <variant>... ; bench stage times of tuning builds (scripts/build_variants.sh)
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for v in "$@"; do
  echo "VARIANT=$v" >> $OUT/variants.txt
  DMR_LIBRARY=$GRAFT_REPO_ROOT/dmesh_renderer_amd/variants/lib_$v.so python bench.py --steps ${STEPS:-30} --warmup 5 --no-early-out --no-tet --stages --no-cpu-baseline 2>$OUT/err_$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('grad_max_norm_err'), d['stages_ms'])" >> $OUT/variants.txt || exit 1
done
cat $OUT/variants.txt
