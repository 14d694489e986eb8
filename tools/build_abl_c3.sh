# experiment: ablated variants of conv3x3.hip (C3_ABL bits: 1 no MFMA, 2 no LDS-DMA, 4 no fragment reads) -> tools/_bin/libavec_c3abl_<n>.so
set -e
cd "$(dirname "$0")/.."
python -m avec_amd.build > /dev/null
OTHERS=$(ls avec_amd/csrc/_obj/*.o | grep -v "/conv3x3.o")
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iavec_amd/csrc -Wno-unused-value -DC3_ABL=$n -c avec_amd/csrc/conv3x3.hip -o tools/_bin/c3_abl_$n.o &
done
wait
for n in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libavec_c3abl_$n.so $OTHERS tools/_bin/c3_abl_$n.o
  rm tools/_bin/c3_abl_$n.o
done
