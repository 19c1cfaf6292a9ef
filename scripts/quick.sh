# usage: bash scripts/quick.sh <name> ; the tri parity tests, then a C4 bench line with the per-stage table (kernel work loop)
set -e
OUT=gpurun_out/$1
mkdir -p $OUT
python -m pytest tests/test_tri_parity_gpu.py tests/test_fuzz_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "not c5 and not tet" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
python bench.py --stages --no-early-out --no-tet --steps 30 > $OUT/bench.json 2> $OUT/bench.err
python - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench.json"))
print(d["value"], "Mpix/s", d["ms_per_step"], "ms/step fwd_err", d.get("fwd_max_abs_err"), "grad_err", d.get("grad_max_norm_err"))
print(d["stages_ms"], "sum", round(sum(d["stages_ms"].values()), 4))
PY
