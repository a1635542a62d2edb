#!/bin/bash
# GP tiers under different workgroup caps (LCFE_GP_GRID_<rows>), serial schedule, gp2d only: ms of the set per pass.
#   tools/gp_grid_sweep.sh 240 "512 384 256 192 128"
NP=$1; shift
for g in $1; do
  v=$(env LCFE_SERIAL=1 LCFE_GP_GRID_$NP=$g python bench.py --sets gp2d --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print(json.load(sys.stdin)['kernel_ms']['gp2d'])")
  echo "tier $NP grid $g: gp2d $v ms"
done
