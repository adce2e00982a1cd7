#!/bin/bash
# A/B (round 5) of the ConvLSTM backward schedules, one gpurun call: parity tests with the default (anti-phase diagonal backward), then the cfg-2 step
#   all three 0                   the serial schedule of round 4
#   SF_LSTM_DIAG_BWD=1            encoder 1's backward on a second stream, its gate kernel released with encoder 2's convolution
#   SF_LSTM_DIAG=1                + the encoder cells' forward diagonal
#   SF_LSTM_WGRAD_SIDE=1          + the decoder cells' weight gradients on the second stream next to the encoder's unroll (all three: the default)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_convlstm_gpu.py tests/test_bf16a_gpu.py tests/test_baseline_size_gpu.py -q -m gpu -x -k "not metnet" 2>&1 | tail -3
for rep in 1 2; do
for v in "SF_LSTM_DIAG=0 SF_LSTM_DIAG_BWD=0 SF_LSTM_WGRAD_SIDE=0" "SF_LSTM_DIAG=0 SF_LSTM_DIAG_BWD=1 SF_LSTM_WGRAD_SIDE=0" "SF_LSTM_DIAG=1 SF_LSTM_DIAG_BWD=1 SF_LSTM_WGRAD_SIDE=0" "SF_LSTM_DIAG=1 SF_LSTM_DIAG_BWD=1 SF_LSTM_WGRAD_SIDE=1"; do
  echo "== $v"
  env $v python bench.py --workload convlstm --steps 40 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms')"
done; done
bash tools/trace_overlap.sh r05_lstm_antiphase SF_LSTM_DIAG=0 SF_LSTM_WGRAD_SIDE=0 -- --workload convlstm --steps 3 --warmup 2 --no-cpu-baseline --no-extra
