set -e
cd tools/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
