#!/bin/bash
set -u
O=gpurun_out/r03p; mkdir -p $O
BBD_BUCKET_BYTES=4000000 timeout 600 python tools/ddp_check.py --capture > $O/capture.log 2>&1; echo "capture rc $?"; grep -v Warning $O/capture.log | tail -15
timeout 900 python -m pytest tests/test_gpu_trainer.py -q -x -k "ranks or rccl or graph" 2>&1 | tail -5
