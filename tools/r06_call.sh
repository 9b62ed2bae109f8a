mkdir -p gpurun_out/r06r
python -m pytest tests/test_gpu_fullres.py -q -x -k "256" > gpurun_out/r06r/fullres.log 2>&1; tail -3 gpurun_out/r06r/fullres.log
for m in 0 1 2; do
IRSDE_TUNING=1 IRSDE_WINO_FUSED64T=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-live-pmc > gpurun_out/r06r/bench_t$m.json 2> gpurun_out/r06r/bench_t$m.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06r/bench_t$m.json").read().strip().splitlines()[-1])
    print("mode $m", d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"))
except Exception as e:
    print("mode $m failed", e); print(open("gpurun_out/r06r/bench_t$m.err").read()[-2000:])
PY
done
