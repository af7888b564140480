set -e
cd vil_fusion_amd/csrc
cp libvilfusion_hip.so /tmp/libvilfusion_hip.prod.so
rm -f vilf_kernels.o
make HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -amdgpu-mfma-vgpr-form=1 -DVILF_STAMPS -Wno-unused-function -Wno-unused-value -Wno-unused-result" > /tmp/make.log 2>&1 || { tail -5 /tmp/make.log; exit 1; }
cd ../..
VILF_DEBUG_STAMPS=1 python tools/dev_prof.py 2048 2>&1 | tail -8
