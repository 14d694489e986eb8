# experiment: ablated variants of the 64-channel weight gradient (C3W_ABL bits: 1 no MFMA, 2 no LDS-DMA, 4 no fragment reads, 8 no atomics) -> tools/_bin/libavec_c3wabl_<n>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
python -m avec_amd.build > /dev/null
OTHERS=$(ls avec_amd/csrc/_obj/*.o | grep -v "/conv3x3.o")
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iavec_amd/csrc -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 -DC3W_ABL=$n $C3W_EXTRA -c avec_amd/csrc/conv3x3.hip -o tools/_bin/c3w_abl_$n.o &
done
wait
for n in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libavec_c3wabl_$n.so $OTHERS tools/_bin/c3w_abl_$n.o
  rm tools/_bin/c3w_abl_$n.o
done
