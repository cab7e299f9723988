set -o pipefail
for g in 1 2 4; do echo "== split $g"; TACO_GL_SPLIT=$g timeout -k 10 120 python tools/gl_bench.py 32 60 || exit 1; TACO_GL_SPLIT=$g timeout -k 10 120 python tools/gl_bench.py 16 60 || exit 1
TACO_GL_SPLIT=$g timeout -k 10 100 python bench.py --to-wav --no-extras --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('wav', d['value'], d['ms_per_step'], d['p50_batch_latency_ms'])" || exit 1
done
