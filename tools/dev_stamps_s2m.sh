# phase stamps of the scan-to-map kernels: diagnostic build of vilf_s2m.hip (make DEFS=-DVILF_STAMPS), one mid-grid workgroup of each launch
set -e
touch vil_fusion_amd/csrc/vilf_s2m.hip vil_fusion_amd/csrc/vilf_kernels.hip vil_fusion_amd/csrc/vilf_marg.hip vil_fusion_amd/csrc/vilf_api.hip
make -s -C vil_fusion_amd/csrc DEFS=-DVILF_STAMPS > /tmp/make.log 2>&1 || { tail -5 /tmp/make.log; exit 1; }
python tools/dev_stamps_s2m.py 2>&1 | tail -12
touch vil_fusion_amd/csrc/vilf_s2m.hip vil_fusion_amd/csrc/vilf_kernels.hip vil_fusion_amd/csrc/vilf_marg.hip vil_fusion_amd/csrc/vilf_api.hip
