#!/bin/bash
# A/B (round 5): the ConvLSTM gate backward taking c' again from the saved bf16 gates (default) against reading it back (SF_LSTM_READ_C=1), alternating on one box.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for cfg in "SF_LSTM_READ_C=1" ""; do
    echo "== ${cfg:-default}"
    env $cfg python bench.py --workload convlstm --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('convlstm %.1f samples/s %.3f ms' % (r['value'], r['ms_per_step']))"
  done
done
