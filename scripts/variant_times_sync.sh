# usage: bash scripts/variant_times_sync.sh <outname> <variant>... ; C4 step with DEFAULT (waiting) calls of tuning builds (scripts/build_variants.sh)
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for v in "$@"; do
  echo "VARIANT=$v" >> $OUT/variants_sync.txt
  DMR_LIBRARY=$GRAFT_REPO_ROOT/dmesh_renderer_amd/variants/lib_$v.so python bench.py --sync --steps 60 --warmup 5 --no-early-out --no-tet --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $OUT/variants_sync.txt || exit 1
done
cat $OUT/variants_sync.txt
