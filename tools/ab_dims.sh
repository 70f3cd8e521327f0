#!/bin/bash
# A/B an alternative library over several dims (development aid): tools/ab_dims.sh <lib> dim:inner ...
lib=$1; shift
for spec in "$@"; do
  dim=${spec%%:*}; inner=${spec#*:}
  for l in "" $lib; do
    PTRWM_LIB=$l PTRWM_KERNEL_FORM=thread python3 bench.py --cpu-seconds 0 --no-extras --steps 6 --warmup 2 --dim $dim --inner $inner > /tmp/o.json 2>/dev/null && python3 -c "import json; d=json.load(open('/tmp/o.json')); print('dim $dim', '${l:-shipped}'.split('/')[-2] if '/' in '${l:-shipped}' else 'shipped', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms'])"
  done
done
