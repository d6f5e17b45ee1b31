#!/bin/bash
# anlmdn A/B on the GPU box: operator parity tests, then the bench line's anlmdn figures
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k anlmdn 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/b_nlm.json 2>gpurun_out/b_nlm.err
python -c "
import json; d=json.load(open('gpurun_out/b_nlm.json')); k=d['second_kernel'] or d['roofline']
print('ms/step', d['ms_per_step'], 'xRT', d['value'], 'nlm ms', k['avg_launch_ms'], k.get('valu'), 'dk ms', d['roofline']['avg_launch_ms'], d['result'])"
