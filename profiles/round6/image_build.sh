#!/bin/bash
# The image build of the depth-14 world on the GPU box's host CPUs, by phase, and the first commit through the library.
#   gpurun -- bash profiles/round6/image_build.sh
set -u
OUT=gpurun_out/r6_image_build; mkdir -p $OUT
T=profiles/round6/tools
CXX=/opt/rocm/lib/llvm/bin/clang++
$CXX -O3 -std=c++17 -pthread -Ivoxel-rs_amd/csrc/hip -Iinclude -o /tmp/imgbench $T/imgbench.cpp
hipcc -O3 -std=c++17 -DVX_PINNED -Ivoxel-rs_amd/csrc/hip -Iinclude -o /tmp/imgbench_pinned $T/imgbench.cpp
g++ -O2 -std=c++17 -pthread -o /tmp/touchbench $T/touchbench.cpp
{
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; ls -d /sys/devices/system/node/node* | wc -l; grep -c processor /proc/cpuinfo; grep Cpus_allowed_list /proc/self/status
/tmp/touchbench 3 16
python3 $T/dump_world.py csvo esvo
for f in "csvo 2" "esvo 1"; do set -- $f
  echo "== $1, the world's bytes in malloc'ed memory"; /tmp/imgbench /tmp/world_$1.bin $2 16 3 1 | grep ok
  echo "== $1, in hipHostMalloc'ed memory (default flags)"; /tmp/imgbench_pinned /tmp/world_$1.bin $2 16 3 1 0 | grep ok
  echo "== $1, in hipHostMalloc'ed memory (hipHostMallocNumaUser)"; /tmp/imgbench_pinned /tmp/world_$1.bin $2 16 3 1 0x20000000 | grep ok
done
python3 profiles/round6/first_commit.py --format csvo --repeats 3 2>&1 | grep -v amdgpu.ids
python3 profiles/round6/first_commit.py --format esvo --repeats 3 2>&1 | grep -v amdgpu.ids
} > $OUT/log.txt 2>&1
cat $OUT/log.txt
