#!/bin/bash
# what each stage costs IN FLIGHT: the bench with a stage taken out (experiments, not the metric)
out=gpurun_out/r3_marg_$1
mkdir -p $out
run() { # name, exp-json
  python bench.py --no-cpu --no-extras --exp "$2" > $out/$1.json 2> $out/$1.err
  python -c "import json; d=json.load(open('$out/$1.json')); print('$1', round(d['value']/1e6,1), round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['stage_ms'].items()})"
}
python bench.py --no-cpu --no-extras > $out/full.json 2> $out/full.err
python -c "import json; d=json.load(open('$out/full.json')); print('full', round(d['value']/1e6,1), round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['stage_ms'].items()})"
run no_stft '{"no_stft": true}'
run no_hp '{"hipass_freq": 0.0}'
run manual '{"on_threshold": 6.0, "off_threshold": 4.0}'
run no_hp_no_stft '{"hipass_freq": 0.0, "no_stft": true}'
run no_hp_manual_no_stft '{"hipass_freq": 0.0, "on_threshold": 6.0, "off_threshold": 4.0, "no_stft": true}'
