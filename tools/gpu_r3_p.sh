#!/bin/bash
# per-dispatch forward durations, one-wave vs two-wave, C3 (same solve: same active counts per step)
set -u
for two in 0 1; do
  TRAJOPT_FWD2=$two bash tools/trace_durations.sh r3p/fwd2_$two quadrotor
done
python - <<'PY'
import re
def rd(p):
    t = open(p).read().split('us:')[1].split()
    return [float(x) for x in t]
a = rd('gpurun_out/r3p/fwd2_0/k_forward.txt'); b = rd('gpurun_out/r3p/fwd2_1/k_forward.txt')
print(len(a), len(b))
print('one:', ' '.join(f'{x:.0f}' for x in a[:150]))
print('two:', ' '.join(f'{x:.0f}' for x in b[:150]))
PY
