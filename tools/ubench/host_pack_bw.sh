#!/bin/bash
# usage (on the GPU box): tools/ubench/host_pack_bw.sh [rows]
set -e
cd "$(dirname "$0")"
g++ -O3 -std=c++17 -c ../../pyfaceanalysis_amd/csrc/hg_hostpack.cpp -o /tmp/hostpack.o
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c host_pack_bw.cpp -o /tmp/host_pack_bw.o
/opt/rocm/bin/hipcc /tmp/host_pack_bw.o /tmp/hostpack.o -o /tmp/host_pack_bw -lpthread
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null || true; lscpu | egrep 'Model name|Socket|NUMA|^CPU\(s\)|Thread' || true
numactl -H 2>/dev/null | head -20 || true
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr "\n" " "; echo "<- numa node of the drm cards"
/tmp/host_pack_bw "$@"
