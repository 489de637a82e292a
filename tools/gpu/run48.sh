#!/bin/bash
# soak: 400 replayed steps -- throughput and memory must not drift
python - <<'PY'
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "400", "--warmup", "10", "--no-secondary", "--no-cpu-baseline", "--no-roofline"], capture_output=True, text=True).stdout
d = json.loads(out.strip().splitlines()[-1])
print("400 steps:", d["value"], d["ms_per_step"], d["step_ms"], d.get("peak_hbm_gb"))
PY
python bench.py --mode frame --steps 60 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('frame 60', d['value'], d['ms_per_step'], d['step_ms'])"
