#!/bin/bash
# One measurement round on the GPU box: parity tests, kernel trace, PMC passes (+ traffic json), full bench line.
# Usage: tools/gpu_round.sh <tag>   -> gpurun_out/<tag>_*  (copy what should be judged into profiles/)
tag=${1:-round}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/${tag}_pytest.txt
cat gpurun_out/${tag}_pytest.txt
tools/gpu_trace.sh ${tag}_trace > /dev/null 2>&1
tools/gpu_pmc.sh ${tag}_pmc > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_pmc --json gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc_summary.txt 2>&1
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic.json
python bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cat gpurun_out/${tag}_bench.json
