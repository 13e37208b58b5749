# one box: the contract's timed region with one and two sub-shard streams, at the driver's flags and at bench.py's default --steps
set -e
Q="--no-other-law --no-cpu-baseline --sustained-steps 0 --no-step-path --no-mixed-set --no-power --no-actor-leg --no-learner-leg --no-facade-leg --no-consumers-leg --no-streaming-point"
run() {
  timeout -k 10 160 python bench.py --gpus 1 "$@" $Q 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', '-> %.1f G' % (d['value'] / 1e9), d['timed_region_phases_us'])"
}
for s in 1 2 1 2 1 2; do run --steps 20 --warmup 5 --streams $s; done
for s in 1 2 1 2; do run --streams $s; done
for s in 1 2 1 2; do run --streams $s --mismatched; done
