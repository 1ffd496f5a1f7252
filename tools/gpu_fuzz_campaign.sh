#!/bin/bash
# A multi-seed run of tools/fuzz_parity.py (every family) + big-image extractions: summary lines into gpurun_out/fuzz_campaign.txt
cd /root/repo
mkdir -p gpurun_out
out=gpurun_out/fuzz_campaign.txt
echo "# tools/fuzz_parity.py, seeds ${1:-201}..$(( ${1:-201} + ${2:-24} - 1 )), 150 cases each (+ families scaled from it), then 6 big-image seeds" > $out
fails=0
for ((s=${1:-201}; s<${1:-201}+${2:-24}; s++)); do
  timeout 600 python tools/fuzz_parity.py --cases 150 --seed $s 2>&1 | grep -E "MISMATCH|# seed|refused" | tail -4 >> $out || fails=$((fails+1))
done
for s in 31 32 33 34 35 36; do
  timeout 600 python tools/fuzz_parity.py --big 10 --seed $s 2>&1 | grep -E "MISMATCH|# seed" | tail -2 >> $out
done
echo "# lines with MISMATCH: $(grep -c MISMATCH $out)" >> $out
tail -12 $out
