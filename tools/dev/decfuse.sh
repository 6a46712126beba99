#!/bin/bash
# developer aid: the fused decode -> scans kernel, before / after / staged (variant libraries, same box)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/decfuse; mkdir -p $O
L=$R/rplidar_ros2_driver_amd/lib
{ for rep in 1 2; do for v in before after staged; do
  unset RPLGPU_DEC_FUSE_STAGE; lib=$L/librplgpu.so
  [ $v = before ] && lib=$L/librplgpu_before.so
  [ $v = staged ] && export RPLGPU_DEC_FUSE_STAGE=1
  echo "== $v"
  for a in 0x85 0x82 0x84 0x86; do RPLGPU_LIBRARY=$lib DEC_ONLY=$a DEC_SUM=1 timeout 200 python tools/dev/decbench.py 4096 2>&1 | grep "segment-fused" | sed "s/^/$a /"; done
done; done; } 2>&1 | tee $O/out.txt
