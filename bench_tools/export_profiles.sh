#!/bin/bash
# Developer tool (this container): gpurun_out/<dir> of collect_profiles.sh -> the tracked summaries under profiles/
#   bench_tools/export_profiles.sh gpurun_out/prof_r02c r02
set -e
D=$1; R=$2
E="python bench_tools/rocpd_export.py"
$E stats $D/stats_headline/runc/*_results.db profiles/${R}_headline_kernel_stats.csv "python bench.py --steps 50 --warmup 5 --no-aux --no-cpu-baseline (64x64 eddy + CGAN, 128 members)"
$E stats $D/stats_config3/runc/*_results.db profiles/${R}_config3_kernel_stats.csv "python bench.py --leg config3 --steps 20 --warmup 5 (96x96 jet + CVAE, 32 members)"
$E stats $D/stats_config4/runc/*_results.db profiles/${R}_config4_kernel_stats.csv "python bench.py --leg config4 --steps 100 --warmup 5 (256x256 unparameterized, 64 members + coarse-grain)"
$E pmc $D/pmcf_headline/runc/*_results.db $D/pmcw_headline/runc/*_results.db profiles/${R}_headline_pmc_hbm_traffic.csv "python bench.py --steps 5 --warmup 2 --no-aux --no-cpu-baseline" profiles/pmc_traffic_f16x3.json "k_convh2<128, 64, 5" '{"nx":64,"members_per_gpu":128,"kind":"gan"}' > /dev/null
$E pmc $D/pmcf_config3/runc/*_results.db $D/pmcw_config3/runc/*_results.db profiles/${R}_config3_pmc_hbm_traffic.csv "python bench.py --leg config3 --steps 5 --warmup 2" profiles/pmc_traffic_config3.json "k_convh2<128, 64, 5" '{"nx":96,"members_per_gpu":32,"kind":"vae"}' > /dev/null
# config4: bytes per STEP over the warm-up + timed steps (5 + 100) of every kernel of the spectral step
$E pmc $D/pmcf_config4/runc/*_results.db $D/pmcw_config4/runc/*_results.db profiles/${R}_config4_pmc_hbm_traffic.csv "python bench.py --leg config4 --steps 100 --warmup 5" profiles/pmc_traffic_config4.json "k_l_team|k_l_cols3|k_l_rows_" '{"nx":256,"members":64,"steps":105}' > /dev/null
ls -la profiles/${R}_*
