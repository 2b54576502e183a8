set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_host_path.py tests/test_gpu_parity.py tests/test_capi_host.py -x -q -m gpu > gpurun_out/r4_gpu_tests_a.log 2>&1 || { tail -40 gpurun_out/r4_gpu_tests_a.log; exit 1; }
tail -3 gpurun_out/r4_gpu_tests_a.log
python tools/host_path.py 16 128 340 728 1738 4096 8192 32768 > gpurun_out/r4_host_path_final.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4_host_path_final.txt | cut -c1-150
