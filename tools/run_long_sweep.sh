python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_h.txt 2>&1; tail -3 gpurun_out/pytest_gpu_h.txt
for cr in 65536 262144 1048576 4194304; do LM_HIP_CHUNK_ROWS=$cr python tools/msweep.py 1000000000 40,64,100 > gpurun_out/msweep_long_chunk$cr.json 2> gpurun_out/msweep_long_chunk$cr.err; done
LM_HIP_CHUNKED_FUSED=0 python tools/msweep.py 1000000000 40,64,100 > gpurun_out/msweep_long_cells.json 2> gpurun_out/msweep_long_cells.err
grep -h call_ms gpurun_out/msweep_long_chunk*.err gpurun_out/msweep_long_cells.err | cut -c1-600
