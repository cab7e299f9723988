set -o pipefail
python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "griffin or gl or wav or no_alloc or cfg3 or smoke or synth" 2>&1 | tail -5 || exit 1
for b in 32 8 1; do timeout -k 10 120 python tools/gl_bench.py $b 60 || exit 1; done
