#!/bin/bash
# Experiment: light curves/s of the default bench under different GP launch plans (LCFE_GP_PLAN / LCFE_GP_STREAMS).
#   tools/gp_plan_sweep.sh "<streams>;<plan>" ...      ("2;" = the default plan)
for spec in "$@"; do
  st=${spec%%;*}; plan=${spec#*;}
  r=$(LCFE_GP_STREAMS=$st LCFE_GP_PLAN="$plan" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1))")
  echo "streams=$st plan='$plan' -> $r" | tee -a gpurun_out/gp_plan_sweep.txt
done
