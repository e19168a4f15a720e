#!/bin/bash
# round 6, GPU job D: the streaming identity form - parity on ragged sizes, timing against the tiled form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "identity_pass_forms" 2>&1 | tail -15
for cfg in md2 boost7; do
  python tools/kernel_bench.py --smooth --iters 200 --config $cfg 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$cfg', 'tiled', d['identity'], {k: v for k, v in d.items() if k.startswith('identity_stream')})" | tee -a gpurun_out/r06/identity_stream_raw.txt
done
