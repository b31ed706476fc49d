#!/bin/bash
# round-2 GPU session 20: no fused skip conv where the 3x3 launch is split over K (one slice at a time): tests + batch-1 bench
set -o pipefail
python -m pytest tests -m gpu -q 2>&1 | tail -3
for i in 1 2; do python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch1', d['value'])"; done
MUD_CONV_SPLITK=0 python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch1, never split (fused skip everywhere)', d['value'])"
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'])"
