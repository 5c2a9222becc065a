#!/bin/bash
# Developer tool (GPU box): the rocprofv3 runs behind profiles/rNN_*: kernel stats, the two HBM PMC passes (FETCH_SIZE,
# WRITE_SIZE in separate runs, MI355X_MICROARCH.md) and one matrix-core pass (SQ_VALU_MFMA_BUSY_CYCLES ...) for the
# headline step and the config3 / config4 legs.  The program itself follows "--" (no env / shell hop).
#   bench_tools/collect_profiles.sh <out dir under gpurun_out/>
set -e
OUT=$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
H="--steps 200 --warmup 20 --no-aux --no-cpu-baseline --no-preheat"
P="--steps 5 --warmup 2 --no-aux --no-cpu-baseline --no-preheat"
M="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16"
rocprofv3 --kernel-trace --stats -d $OUT/stats_headline -- python bench.py $H > $OUT/headline.json 2> /dev/null
echo headline stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmcf_headline -- python bench.py $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmcw_headline -- python bench.py $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $M -d $OUT/pmcm_headline -- python bench.py $P > /dev/null 2>&1
echo headline pmc done
# the legs run their own fixed protocols (config3: 10 + 200 steps; config4: 24 + 1000 steps, diagnostics cadence in the
# second half, one coarse-grain)
# (config3's PMC passes with --one-stream: whole-ensemble launches only, one kernel shape per name; its kernel stats as the leg runs)
for leg in config3 config4; do
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$leg -- python bench.py --leg $leg > $OUT/$leg.json 2> /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmcf_$leg -- python bench.py --leg $leg --one-stream > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmcw_$leg -- python bench.py --leg $leg --one-stream > /dev/null 2>&1
  echo $leg done
done
rocprofv3 --kernel-trace --pmc $M -d $OUT/pmcm_config3 -- python bench.py --leg config3 --one-stream > /dev/null 2>&1
ls $OUT
