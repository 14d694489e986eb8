# build a variant library tools/_bin/libavec_<name>.so from the CURRENT working tree (for same-box A/B runs via AVEC_LIB_PATH): bash tools/build_variant.sh <name>
set -e
cd "$(dirname "$0")/.."
python -m avec_amd.build > /dev/null
mkdir -p tools/_bin
cp avec_amd/libavec_hip.so tools/_bin/libavec_$1.so
ls -la tools/_bin/libavec_$1.so
