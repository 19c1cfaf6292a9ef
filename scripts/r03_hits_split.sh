# k_tri_backward_hits with 1..4 workgroups per tile: stage times (bench.py's separate all-stage pass) at C4 full / halves / eighths, C2, C1
set -e
OUT=gpurun_out/hs; mkdir -p $OUT
for S in 1 2 3 4; do
  for spec in C4:0/1 C4:0/2 C4:0/4 C4:0/8 C2:0/1 C1:0/1 C5:0/8; do
    cfg=${spec%%:*}; rk=${spec##*:}; tag=${cfg}_$(echo $rk | tr / _)_s$S
    steps=30; if [ "$cfg" = "C5" ]; then steps=8; fi
    DMR_HITS_SPLIT=$S timeout -k 10 300 python bench.py --config $cfg --emulate-rank $rk --partition bands --steps $steps --warmup 3 --no-cpu-baseline > $OUT/$tag.json
    python - $OUT/$tag.json $tag <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "ms/step", j["ms_per_step"], "hits", j["stages_ms"].get("k_tri_backward_hits"), "pix", j["stages_ms"].get("k_tri_backward_pix"), "fwd", j["stages_ms"].get("k_tri_forward"), flush=True)
PY
  done
done
