# usage: bash scripts/variant_times_tet.sh <outname> <variant>... ; C3 (tet) stage times of tuning builds (scripts/build_variants.sh)
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for v in "$@"; do
  echo "VARIANT=$v" >> $OUT/variants_tet.txt
  DMR_LIBRARY=$GRAFT_REPO_ROOT/dmesh_renderer_amd/variants/lib_$v.so python bench.py --config C3 --steps 30 --warmup 5 --stages --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms'])" >> $OUT/variants_tet.txt || exit 1
done
cat $OUT/variants_tet.txt
