mkdir -p gpurun_out
python tools/form_sweep.py dense > gpurun_out/r03_form_dense.txt 2>&1; tail -2 gpurun_out/r03_form_dense.txt
