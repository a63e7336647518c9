timeout -k 10 900 python -m pytest tests -q -x -m gpu > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
for i in 1 2; do python bench.py --workload iaog --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('iaog', d['value'], d['ms_per_step'])"; done
python bench.py --workload iaog --batch 16 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('iaog B16', d['value'], d['ms_per_step'])"
python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fcmf', d['value'], d['ms_per_step'])"
