#!/bin/bash
# Developer tool (this container): gpurun_out/<dir> of collect_profiles.sh -> the tracked summaries under profiles/
#   bench_tools/export_profiles.sh gpurun_out/prof_r03 r03
set -e
D=$1; R=$2
E="python bench_tools/rocpd_export.py"
db() { ls $1/*/*_results.db | head -1; }
$E stats $(db $D/stats_headline) profiles/${R}_headline_kernel_stats.csv "python bench.py --steps 200 --warmup 20 --no-aux --no-cpu-baseline --no-preheat (64x64 eddy + CGAN, 128 members: 220 steps, no pre-heat launches)"
$E stats $(db $D/stats_config3) profiles/${R}_config3_kernel_stats.csv "python bench.py --leg config3 (96x96 jet + CVAE, 32 members, 10 + 200 steps)"
$E stats $(db $D/stats_config4) profiles/${R}_config4_kernel_stats.csv "python bench.py --leg config4 (256x256 unparameterized, 64 members, 24 + 1000 steps, diagnostics cadence in the second half, one coarse-grain)"
$E pmc $(db $D/pmcf_headline) $(db $D/pmcw_headline) profiles/${R}_headline_pmc_hbm_traffic.csv "python bench.py --steps 5 --warmup 2 --no-aux --no-cpu-baseline --no-preheat" profiles/pmc_traffic_f16x3.json "k_convw2<64" '{"nx":64,"members_per_gpu":128,"members_per_launch":128,"kind":"gan"}' > /dev/null
$E pmc $(db $D/pmcf_config3) $(db $D/pmcw_config3) profiles/${R}_config3_pmc_hbm_traffic.csv "python bench.py --leg config3 --one-stream" profiles/pmc_traffic_config3.json "k_convw2<96" '{"nx":96,"members_per_gpu":32,"kind":"vae"}' > /dev/null
# config4: bytes per STEP over the warm-up + timed steps (24 + 1000) of every kernel of the spectral step, the diagnostics
# increments' transforms (21 of them) included
$E pmc $(db $D/pmcf_config4) $(db $D/pmcw_config4) profiles/${R}_config4_pmc_hbm_traffic.csv "python bench.py --leg config4" profiles/pmc_traffic_config4.json "k_l_team|k_l_cols|k_l_rows_|k_diag" '{"nx":256,"members":64,"steps":1024}' > /dev/null
$E mfma $(db $D/pmcm_headline) profiles/${R}_headline_pmc_mfma.csv "python bench.py --steps 5 --warmup 2 --no-aux --no-cpu-baseline --no-preheat (B=128, N=64, GAN, f16x3)"
$E mfma $(db $D/pmcm_config3) profiles/${R}_config3_pmc_mfma.csv "python bench.py --leg config3 --one-stream (96x96 jet + CVAE, 32 members)"
ls -la profiles/${R}_*
