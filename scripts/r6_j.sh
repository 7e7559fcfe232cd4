cd /root/repo
bash scripts/collect_profile.sh r06_a
grep -A6 "trainer kernels" gpurun_out/r06_a_summary.txt
grep -B1 -A10 "maf_chain_kernel<16, 4, false, 0>  grid" gpurun_out/r06_a_summary.txt | head -60
bash scripts/collect_round.sh r06_a
tail -5 gpurun_out/r06_a/gpu_tests.txt
