# experiment: stage-1 slab kernel with in-kernel phase stamps -> tools/_bin/libavec_c3strace.so  (bash tools/build_trace_c3s.sh [workgroup])
set -e
cd "$(dirname "$0")/.."
python -m avec_amd.build > /dev/null
mkdir -p tools/_bin
OTHERS=$(ls avec_amd/csrc/_obj/*.o | grep -v "/conv3x3.o")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iavec_amd/csrc -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 -DC3S_TRACE -DC3S_TRACE_WG=${1:-5} -c avec_amd/csrc/conv3x3.hip -o tools/_bin/c3s_trace.o
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libavec_c3strace.so $OTHERS tools/_bin/c3s_trace.o
rm tools/_bin/c3s_trace.o
