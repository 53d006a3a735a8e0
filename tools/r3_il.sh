#!/bin/bash
# tracker on the interleaved envelope (OFP_MM_IL): parity subset, the in-flight bench with and without, the lone call
export OFP_MM_IL=1
timeout -k 10 800 python -m pytest tests/test_gpu_detect.py tests/test_gpu_pipeline.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/il_tests.log 2>&1 || { tail -30 gpurun_out/il_tests.log; exit 1; }
tail -3 gpurun_out/il_tests.log
bash tools/r3_il2.sh
OFP_MM_IL=1 bash tools/alone_trace.sh il1 | head -18
