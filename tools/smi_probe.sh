#!/bin/bash
# Samples rocm-smi (sclk / power / temperature) once a second WHILE bench.py runs its timed steps: is the step clock- or
# power-capped?  usage (GPU box): bash tools/smi_probe.sh [extra bench flags]  ->  gpurun_out/smi_probe/{bench.json, smi.txt}
OUT=gpurun_out/smi_probe
mkdir -p $OUT
python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 800 "$@" > $OUT/bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 40); do
    if ! kill -0 $BP 2>/dev/null; then break; fi
    echo "t=$i $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|Power|junction' | tr -s ' \t' ' ' | tr '\n' '|')" >> $OUT/smi.txt
    sleep 1
done
wait $BP
cat $OUT/smi.txt
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('frames/s', round(d['value']), 'ms/step', round(d['ms_per_step'],2))"
