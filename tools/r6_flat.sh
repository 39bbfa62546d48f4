#!/bin/bash
# Round 6: the shadow passes after a change of h16_stream -- parity (FLAT + IVF shadow paths), then the exhaustive pass and the headline step
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "flat or candidate_pass or shadow or prun or mfma" --timeout 900 2>&1 | tail -4
FLAT_CHECK=1 python tools/flat_batch.py 5 4096 "$@" 2>&1 | grep -v "^W2026\|amdgpu.ids"
python bench.py --headline-only --steps 20 --warmup 5 --no-concurrent 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'])"
