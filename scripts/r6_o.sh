cd /root/repo
bash scripts/collect_profile.sh r06_b
bash scripts/collect_profile.sh r06_c --no-flow-bench
grep -A10 "trainer kernels" gpurun_out/r06_c_summary.txt | head -14
bash scripts/collect_round.sh r06_b
tail -5 gpurun_out/r06_b/gpu_tests.txt
python scripts/readme_fit_share.py /root/repo nsf6 2>/dev/null | tail -1 > gpurun_out/r06_b/readme_after.json
cat gpurun_out/r06_b/readme_after.json
