#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4h; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
{
for n in 1 2 3 4; do echo "== S workgroups per CU: $n"; RPLGPU_VOXEL_PATH=two RPLGPU_VOXEL_PIPE=$n timeout 120 python tools/dev/pipetl.py 2>&1 | tail -4; RPLGPU_VOXEL_PIPE=$n timeout 120 python tools/dev/vbench.py 4096 20 2>&1 | tail -1; done
for n in 3 4; do echo "== c8 (64-register consumer) S workgroups per CU: $n"; RPLGPU_LIBRARY=$LIB/librplgpu_c8.so RPLGPU_VOXEL_PATH=two RPLGPU_VOXEL_PIPE=$n timeout 120 python tools/dev/pipetl.py 2>&1 | tail -4;  RPLGPU_LIBRARY=$LIB/librplgpu_c8.so RPLGPU_VOXEL_PIPE=$n timeout 120 python tools/dev/vbench.py 4096 20 2>&1 | tail -1; done
} 2>&1 | tee $O/occ.txt
