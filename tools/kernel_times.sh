#!/bin/bash
# Per-kernel serial times (us) of the fit sets under rocprofv3 for the library in LCFE_LIB_PATH (default: the tree's):
#   tools/kernel_times.sh <tag> [sets]
TAG=$1; SETS=${2:-bazin,powerlaw}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LCFE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sets $SETS > /dev/null 2> gpurun_out/kt_$TAG.err
f=$(find gpurun_out/kt_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$TAG" <<'PY'
import csv,re,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"\(.*","",n).replace("void ","")
    if float(r["AverageNs"])>2e6: print(f'{sys.argv[2]:8s} {n:40s} {float(r["AverageNs"])/1e3:10.0f} us')
PY
