cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3e
python -m pytest tests/test_kernels_gpu.py -q -x -k nt256 2>&1 | tail -5 | cut -c1-300
python tools/gemm_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3e/sweep.txt; cat gpurun_out/r3e/sweep.txt
python tools/gemm_bench.py --model base 2>&1 | grep -v amdgpu.ids > gpurun_out/r3e/gemm_base.txt; cat gpurun_out/r3e/gemm_base.txt
python tools/gemm_bench.py --model small 2>&1 | grep -v amdgpu.ids > gpurun_out/r3e/gemm_small.txt; cat gpurun_out/r3e/gemm_small.txt
