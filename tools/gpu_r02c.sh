#!/bin/bash
# micro-benchmarks of the ICP reduction alone: launch-shape sweep and ablations, this build vs the round-1 build (tools/ab/)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02c}
mkdir -p $O
cd $R
echo "== sweep, this build"; timeout 120 python tools/icp_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/sweep_new.txt
echo "== sweep, round-1 build"; CF_HIP_LIB=$R/tools/ab/libcofusion_hip_r01.so timeout 120 python tools/icp_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/sweep_r01.txt
for shape in "256 1" "256 4"; do for ab in 0 1 2 4; do
  echo -n "new shape $shape ablate $ab: "; CF_ICP_ABLATE=$ab timeout 60 python tools/icp_one.py $shape 300 2>&1 | grep "L0 us"
done; done | tee $O/ablate.txt
for ab in 0 1 2 4; do echo -n "r01 shape 256 1 ablate $ab: "; CF_HIP_LIB=$R/tools/ab/libcofusion_hip_r01.so CF_ICP_ABLATE=$ab timeout 60 python tools/icp_one.py 256 1 300 2>&1 | grep "L0 us"; done | tee -a $O/ablate.txt
