#!/bin/bash
# A/B of the diagonal (two-stream) encoder schedule of the ConvLSTM stack, one gpurun call: parity tests with the switch on, then the cfg-2 step
# with and without it, with the default 8-wave cell kernel and with the two-workgroups-per-CU variant (SF_LSTM_W4=1 SF_LSTM_WS=1).
cd $GRAFT_REPO_ROOT
SF_LSTM_DIAG=1 python -m pytest tests/test_convlstm_gpu.py tests/test_bf16a_gpu.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2; do
for v in "SF_LSTM_DIAG=0" "SF_LSTM_DIAG=1" "SF_LSTM_DIAG=0 SF_LSTM_W4=1 SF_LSTM_WS=1" "SF_LSTM_DIAG=1 SF_LSTM_W4=1 SF_LSTM_WS=1"; do
  echo "== $v"
  env $v python bench.py --workload convlstm --steps 40 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms')"
done; done
