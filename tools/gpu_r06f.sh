# round 6, last: a wider fuzz campaign on the final library (six seeds x 120 cases, one contention campaign)
export TMPDIR=/tmp
mkdir -p gpurun_out
for seed in 6301 6302 6303 6304 6305 6306; do
  timeout 600 python tools/fuzz_parity.py --cases 120 --seed $seed > /tmp/fz_$seed.txt 2>&1; echo "seed $seed rc=$? $(tail -1 /tmp/fz_$seed.txt | cut -c1-160)"
done | tee gpurun_out/r06aw_fuzz_campaign.txt
timeout 900 python tools/fuzz_parity.py --contention 40 --seed 6310 2>&1 | tail -2 | tee -a gpurun_out/r06aw_fuzz_campaign.txt
