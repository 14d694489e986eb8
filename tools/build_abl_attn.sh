# experiment: ablated variants of the MFMA attention kernels (AVEC_ATTN_ABL: 1 = return after staging, 2 = after the scores / softmax, 3 (backward) = after the dP / dS loop) -> tools/_bin/libavec_attn_abl_<n>.so
set -e
cd "$(dirname "$0")/.."
python -m avec_amd.build > /dev/null
mkdir -p tools/_bin
OTHERS=$(ls avec_amd/csrc/_obj/*.o | grep -v "/attention_mfma.o")
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iavec_amd/csrc -Wno-unused-value -DAVEC_ATTN_ABL=$n -c avec_amd/csrc/attention_mfma.hip -o tools/_bin/attn_abl_$n.o &
done
wait
for n in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libavec_attn_abl_$n.so $OTHERS tools/_bin/attn_abl_$n.o
  rm tools/_bin/attn_abl_$n.o
done
ls -la tools/_bin/
