for form in thread quad; do
for inner in 1 2 5 10; do
PTRWM_KERNEL_FORM=$form python3 bench.py --cpu-seconds 0 --no-extras --steps 200 --warmup 20 --inner $inner > gpurun_out/i1_${form}_$inner.json 2>/dev/null
python3 -c "import json; d=json.load(open('gpurun_out/i1_${form}_$inner.json')); b=2*65536*32*(30*4+4+24+8); print('$form', $inner, '%.4g steps/s' % d['value'], '%.4f ms' % d['roofline']['kernel_ms'], '%.2f TB/s' % (b/d['roofline']['kernel_ms']/1e9))"
done; done
