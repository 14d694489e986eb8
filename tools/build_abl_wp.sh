# experiment: ablated variants of the pair weight-gradient kernel (WP_ABL bits: 1 no LDS-DMA after the first stage, 2 no MFMA, 4 no final atomics, 8 no row masks) -> tools/_bin/libavec_wp_abl_<n>.so
set -e
cd "$(dirname "$0")/.."
python -m avec_amd.build > /dev/null
mkdir -p tools/_bin
OTHERS=$(ls avec_amd/csrc/_obj/*.o | grep -v "/wgrad_pairs.o")
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iavec_amd/csrc -Wno-unused-value -DWP_ABL=$n -c avec_amd/csrc/wgrad_pairs.hip -o tools/_bin/wp_abl_$n.o &
done
wait
for n in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libavec_wp_abl_$n.so $OTHERS tools/_bin/wp_abl_$n.o
  rm tools/_bin/wp_abl_$n.o
done
