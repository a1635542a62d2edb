#!/bin/bash
# Default (concurrent) schedule under workgroup caps of the GP tiers: light curves/s of the whole step.
#   tools/schedule_sweep.sh "LCFE_GP_GRID_240=256" "LCFE_GP_GRID_240=256 LCFE_GP_GRID_112=512" ...
run() {
  v=$(env $1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), round(d['ms_per_step'],1), round(d['kernel_ms']['gp2d'],1))")
  echo "[$1] -> $v (lc/s, ms/step, gp2d event ms)"
}
run "LCFE_NONE=1"
for cfg in "$@"; do run "$cfg"; done
