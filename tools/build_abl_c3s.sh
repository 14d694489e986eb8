# experiment: ablated variants of the stage-1 slab kernel (C3S_ABL bits: see csrc/conv3x3.hip) -> tools/_bin/libavec_c3sabl_<n>.so
set -e
cd "$(dirname "$0")/.."
python -m avec_amd.build > /dev/null
mkdir -p tools/_bin
OTHERS=$(ls avec_amd/csrc/_obj/*.o | grep -v "/conv3x3.o")
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iavec_amd/csrc -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 -DC3S_ABL=$n -c avec_amd/csrc/conv3x3.hip -o tools/_bin/c3s_abl_$n.o &
done
wait
for n in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libavec_c3sabl_$n.so $OTHERS tools/_bin/c3s_abl_$n.o
  rm tools/_bin/c3s_abl_$n.o
done
