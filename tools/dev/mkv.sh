#!/bin/bash
# tools/dev/mkv.sh <tag> [-D flags...] — developer aid: lib/librplgpu_<tag>.so = the tree's objects
# (make) with rpl_voxel.hip recompiled under the given flags, for same-box A/B runs
# (RPLGPU_LIBRARY=.../librplgpu_<tag>.so).
set -e
R=$(cd $(dirname $0)/../.. && pwd); C=$R/rplidar_ros2_driver_amd/csrc; O=$R/build/obj
TAG=$1; shift
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I$R/include -I$C"
/opt/rocm/bin/hipcc $FL "$@" -c $C/rpl_voxel.hip -o $O/rpl_voxel_$TAG.o
OBJS=$(ls $O/rpl_kernels.o $O/rpl_laserscan.o $O/rpl_ror.o $O/rpl_decode.o $O/rpl_msg.o $O/rpl_fuse.o $O/rpl_project.o $O/rpl_comm.o $O/rplgpu_api.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/rplidar_ros2_driver_amd/lib/librplgpu_$TAG.so $OBJS $O/rpl_voxel_$TAG.o -ldl
echo built librplgpu_$TAG.so
