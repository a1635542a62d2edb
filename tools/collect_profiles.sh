#!/bin/bash
# Collect the rocprofv3 summaries of a round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r03
# kernel-trace/stats and the PMC counters are collected in SEPARATE rocprofv3 runs (the pool refuses combining them).
set -o pipefail
R=${1:-r03}
O=gpurun_out/prof_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
copy_csv() { find "$1" -name "*$2.csv" | head -1 | xargs -I{} cp {} "$3"; }
# (a) serial per-kernel durations, (b) the default (concurrent) schedule
LCFE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_bench_serial.json 2> $O/serial.err && copy_csv $O/serial kernel_stats $O/${R}_bench_serial_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/default -- python3 bench.py --steps 3 --warmup 1 > $O/${R}_bench_default.json 2> $O/default.err && copy_csv $O/default kernel_stats $O/${R}_bench_default_kernel_stats.csv
# (c) statistics set: HBM traffic and instruction counters, one counter group per pass
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/stat_$c -- python3 bench.py --sets stat --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/stat_$c.err && copy_csv $O/stat_$c counter_collection $O/${R}_stat_pmc_$(echo $c | tr A-Z a-z).csv
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/stat_sq -- python3 bench.py --sets stat --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/stat_sq.err && copy_csv $O/stat_sq counter_collection $O/${R}_stat_pmc_sq.csv
# (d) GP tiers: fabric traffic of a 20,000-object batch
for c in FETCH_SIZE WRITE_SIZE; do
  LCFE_SERIAL=1 rocprofv3 --pmc $c --output-format csv -d $O/gp_$c -- python3 bench.py --sets gp2d --steps 1 --warmup 0 --no-cpu-baseline --objects 20000 > /dev/null 2> $O/gp_$c.err && copy_csv $O/gp_$c counter_collection $O/${R}_gp_pmc_$(echo $c | tr A-Z a-z).csv
done
# (e) fit and GP kernels: VALU / MFMA occupancy and wait counters on a 20,000-object batch
LCFE_SERIAL=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $O/fits_sq -- python3 bench.py --sets bazin,powerlaw --steps 1 --warmup 0 --no-cpu-baseline --objects 20000 > /dev/null 2> $O/fits_sq.err && copy_csv $O/fits_sq counter_collection $O/${R}_fits_pmc_sq.csv
LCFE_SERIAL=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $O/gp_sq -- python3 bench.py --sets gp2d --steps 1 --warmup 0 --no-cpu-baseline --objects 20000 > /dev/null 2> $O/gp_sq.err && copy_csv $O/gp_sq counter_collection $O/${R}_gp_pmc_sq.csv
ls -la $O/*.csv $O/*.json
