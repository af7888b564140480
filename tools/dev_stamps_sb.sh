set -e
cd vil_fusion_amd/csrc
rm -f vilf_kernels.o
make HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -amdgpu-mfma-vgpr-form=1 -DVILF_STAMPS -Wno-unused-function -Wno-unused-value -Wno-unused-result" > /tmp/make.log 2>&1 || { tail -5 /tmp/make.log; exit 1; }
cd ../..
for B in 1 256 4096; do VILF_DEBUG_STAMPS=1 python tools/dev_prof.py $B 2>&1 | grep -v "^linearize\|^step\|^schur\|^cholesky\|^solve phase" | tail -5; done
