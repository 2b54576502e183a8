set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_host_path.py tests/test_capi_host.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4_hp_tests.log 2>&1 || { tail -30 gpurun_out/r4_hp_tests.log; exit 1; }
tail -3 gpurun_out/r4_hp_tests.log
python tools/host_path.py 16 128 340 728 1738 4096 > gpurun_out/r4_host_path_final2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4_host_path_final2.txt | cut -c1-150
HIGSFA_HOST_DIRECT=0 python tools/host_path.py 16 728 4096 2>&1 | grep -v amdgpu.ids | cut -c1-150
