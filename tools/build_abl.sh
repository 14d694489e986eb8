# experiment: ablated variants of the NT GEMM kernel (AVEC_ABL bits: 1 no MFMA, 2 no LDS-DMA, 4 no fragment reads, 8 no barrier) -> tools/_bin/libavec_abl_<n>.so
set -e
cd "$(dirname "$0")/.."
python -m avec_amd.build > /dev/null
OTHERS=$(ls avec_amd/csrc/_obj/*.o | grep -v "/gemm.o")
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iavec_amd/csrc -Wno-unused-value -DAVEC_ABL=$n -c avec_amd/csrc/gemm.hip -o tools/_bin/gemm_abl_$n.o &
done
wait
for n in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libavec_abl_$n.so $OTHERS tools/_bin/gemm_abl_$n.o
  rm tools/_bin/gemm_abl_$n.o
done
ls -la tools/_bin/
