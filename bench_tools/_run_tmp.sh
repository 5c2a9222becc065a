set -e
python -m pytest tests/test_gpu_diagnostics.py tests/test_gpu_large_team.py tests/test_gpu_statistics.py -x -q -k "diag or cadence or time_averaged or checksums" > gpurun_out/r3_gputest4.log 2>&1 || true
tail -5 gpurun_out/r3_gputest4.log
python bench.py --leg config4 > gpurun_out/r3_c4.json 2>&1
tail -c 1500 gpurun_out/r3_c4.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b1 -- python bench.py --leg b1 > gpurun_out/r3_b1_prof.json 2>/dev/null
python bench_tools/rocpd_export.py stats gpurun_out/prof_b1/*/*_results.db gpurun_out/r3_b1_kernel_stats.csv "bench.py --leg b1"
cut -c1-150 gpurun_out/r3_b1_kernel_stats.csv | head -24
