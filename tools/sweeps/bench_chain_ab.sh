# headline + configs[4] leg with the fp16-only chain on / off (run through gpurun from the repo root)
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact > gpurun_out/b2.json 2> gpurun_out/b2.err
python - <<PY
import json
d=json.loads(open("gpurun_out/b2.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: d["config4_fp16"][k] for k in ("value", "ms_per_step")}, d.get("round1_workload"))
PY
JTSM_CHAIN16=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact --no-roofline > gpurun_out/b3.json 2> gpurun_out/b3.err
python - <<PY
import json
d=json.loads(open("gpurun_out/b3.json").read().strip().splitlines()[-1])
print("chain off:", d["value"], d["ms_per_step"], {k: d["config4_fp16"][k] for k in ("value", "ms_per_step")})
PY
