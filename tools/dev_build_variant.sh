# build a variant of the library from the working tree into tools/ab/<name>.so: bash tools/dev_build_variant.sh <name> [DEFS...]
set -e
N=$1; shift
D=/tmp/vb_$N
rm -rf $D && mkdir -p $D/vil_fusion_amd/csrc && cp -r include $D/ && cp vil_fusion_amd/csrc/*.hip vil_fusion_amd/csrc/*.hpp vil_fusion_amd/csrc/Makefile $D/vil_fusion_amd/csrc/
make -C $D/vil_fusion_amd/csrc -j8 DEFS="$*" 2>&1 | grep -E "error" -A5 | head -20 || true
mkdir -p tools/ab && cp $D/vil_fusion_amd/csrc/libvilfusion_hip.so tools/ab/$N.so && ls -la tools/ab/$N.so
