#!/bin/bash
# Developer tool (GPU box): the rocprofv3 runs behind profiles/rNN_*: kernel stats and the two HBM PMC passes
# (FETCH_SIZE, WRITE_SIZE in separate runs, MI355X_MICROARCH.md) for the headline step and the config3 / config4 legs.
#   bench_tools/collect_profiles.sh <out dir under gpurun_out/>
set -e
OUT=$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
H="--steps 50 --warmup 5 --no-aux --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats_headline -- python bench.py $H > $OUT/headline.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmcf_headline -- python bench.py --steps 5 --warmup 2 --no-aux --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmcw_headline -- python bench.py --steps 5 --warmup 2 --no-aux --no-cpu-baseline > /dev/null 2>&1
# config4 runs its steps as ONE run of the XCD-resident kernel per call: the passes use the leg's own step count
for leg in config3 config4; do
  if [ $leg = config4 ]; then S="--steps 100 --warmup 5"; P="--steps 100 --warmup 5"; else S="--steps 20 --warmup 5"; P="--steps 5 --warmup 2"; fi
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$leg -- python bench.py --leg $leg $S > $OUT/$leg.json 2> /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmcf_$leg -- python bench.py --leg $leg $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmcw_$leg -- python bench.py --leg $leg $P > /dev/null 2>&1
done
ls $OUT
