# same-box A/B of b_map_update's workgroup size (MU_T): rebuilds vilf_s2m.o per variant, runs the LiDAR-only bench, prints the voxel-grid group
set -e
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-function -Wno-unused-value -Wno-unused-result"
for T in 1024 256 512 256 1024; do
  (cd vil_fusion_amd/csrc && /opt/rocm/bin/hipcc $FL -DMU_T=$T -c vilf_s2m.hip -o vilf_s2m.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libvilfusion_hip.so vilf_kernels.o vilf_marg.o vilf_s2m.o vilf_feat.o vilf_api.o vilf_host.o vilf_init.o vilf_pg.o vilf_lw.o vilf_comm.o -lpthread -ldl)
  python bench.py --distinct-lidar 8 --no-cpu-baseline --no-pcie --ragged-windows 0 --converging-windows 0 --steps 8 > /tmp/ab.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/ab.json')); k=d['roofline']['kernels_ms_per_step']; print('MU_T', $T, 'voxel_grid', round(k['s2m_voxel_grid'],3), 'step', round(d['ms_per_step'],2))"
done
