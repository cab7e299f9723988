set -o pipefail
for st in 3 4 6 8; do
timeout -k 10 100 python bench.py --to-wav --streams $st --no-extras --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('wav streams $st', d['value'], d['ms_per_step'])" || exit 1
done
for st in 6 8; do
timeout -k 10 100 python bench.py --streams $st --no-extras --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('mel streams $st', d['value'], d['ms_per_step'])" || exit 1
done
export GPU_MAX_HW_QUEUES=16
for st in 6 8; do
timeout -k 10 100 python bench.py --keep-hw-queues --streams $st --no-extras --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('mel hwq16 streams $st', d['value'], d['ms_per_step'])" || exit 1
done
timeout -k 10 100 python bench.py --keep-hw-queues --to-wav --streams 6 --no-extras --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('wav hwq16 streams 6', d['value'], d['ms_per_step'])" || exit 1
