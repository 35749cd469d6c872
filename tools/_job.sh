python -m pytest tests -m gpu -x -q > gpurun_out/r3_t3.log 2>&1; tail -5 gpurun_out/r3_t3.log
python - > gpurun_out/r3_dropin.txt 2>&1 <<'PY'
import json, torch, bench, ntracer_amd
from ntracer_amd import tracern
print(json.dumps(bench.dropin_render(torch, ntracer_amd, tracern), indent=1))
PY
cat gpurun_out/r3_dropin.txt
python tools/il_ab.py --cases head --steps 40 --rounds 4
python tools/il_ab.py --cases head --steps 5 --rounds 8
