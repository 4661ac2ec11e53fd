#!/bin/bash
# interleaved A/B of one environment switch on the headline and the grounding stream:  tools/probes/ab_env.sh MMDUET_SOMETHING [rounds]
V=$1; R=${2:-2}
for i in $(seq $R); do
  for cfg in stream300 ground600; do
    for on in 0 1; do
      if [ $on = 1 ]; then export $V=1; else unset $V; fi
      python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --multi-stream 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg $V=$on', d['value'], d['ms_per_step'], d['verified']['steps_bit_identical'])"
    done
  done
done
