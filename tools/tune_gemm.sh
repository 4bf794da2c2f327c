#!/bin/bash
# PyTorch TunableOp search over the library GEMMs of the headline workload (hipBLASLt / rocBLAS solutions per shape),
# then an A/B of the bench with and without the tuned table.  On the GPU box:  bash tools/tune_gemm.sh <batch>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=${1:-128}
cd $R
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$R/gpurun_out/tunableop_b$B.csv \
       PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=20 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=3 PYTORCH_TUNABLEOP_VERBOSE=0
date
timeout -k 10 1000 python bench.py --batch $B --no-cpu-baseline --no-roofline --no-also --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-200
date
ls -la gpurun_out/tunableop_b$B*.csv; wc -l gpurun_out/tunableop_b$B*.csv
export PYTORCH_TUNABLEOP_TUNING=0
for i in 1 2; do
  echo "== tuned";   python bench.py --batch $B --no-cpu-baseline --no-roofline --no-also 2>/dev/null | cut -c150-260
  echo "== untuned"; PYTORCH_TUNABLEOP_ENABLED=0 python bench.py --batch $B --no-cpu-baseline --no-roofline --no-also 2>/dev/null | cut -c150-260
done
