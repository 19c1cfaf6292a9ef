# usage: bash scripts/gpu_check.sh <name> ; GPU parity tests + a bench line with the per-stage table
set -e
OUT=gpurun_out/$1
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
python bench.py --stages > $OUT/bench.json 2> $OUT/bench.err
python - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench.json"))
print(d["value"], "Mpix/s", d["ms_per_step"], "ms/step fwd_err", d.get("fwd_max_abs_err"), "grad_err", d.get("grad_max_abs_err"))
print(d["stages_ms"])
PY
