cd /root/repo
mkdir -p gpurun_out/r6i
python scripts/readme_fit_share.py /root/repo/scripts/abl/r05tree nsf6 2>/dev/null | tail -1 > gpurun_out/r6i/readme_before.json
python scripts/readme_fit_share.py /root/repo nsf6 2>/dev/null | tail -1 > gpurun_out/r6i/readme_after.json
python scripts/readme_fit_share.py /root/repo/scripts/abl/r05tree maf3 2>/dev/null | tail -1 > gpurun_out/r6i/readme_before_maf3.json
python scripts/readme_fit_share.py /root/repo maf3 2>/dev/null | tail -1 > gpurun_out/r6i/readme_after_maf3.json
cat gpurun_out/r6i/readme_*.json
bash scripts/collect_profile.sh r06_a
head -40 gpurun_out/r06_a_summary.txt
