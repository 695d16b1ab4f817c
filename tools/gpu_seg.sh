#!/bin/bash
# segmentation kernels: parity tests, then the configs[2] bench with the in-kernel phase clocks (CF_SEG_CLOCK builds only)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-seg}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_facade_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log
CF_SEG_CLOCK=1 timeout 200 python bench.py --workload objects4 --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tail -4
timeout 200 python bench.py --workload objects8 --steps 60 --warmup 20 --no-cpu-baseline --no-extras 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tail -1
