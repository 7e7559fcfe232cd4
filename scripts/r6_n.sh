cd /root/repo
for b in 64 512; do
python scripts/time_fit.py 32 3 0 5120 $b f32 2>/dev/null | tail -1
python scripts/time_fit.py 128 8 512 5120 $b f32 2>/dev/null | tail -1
python scripts/time_fit.py 128 8 512 5120 $b bf16 2>/dev/null | tail -1
done
