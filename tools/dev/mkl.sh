#!/bin/bash
# tools/dev/mkl.sh <tag> [-D flags...] — as mkv.sh, for rpl_laserscan.hip
set -e
R=$(cd $(dirname $0)/../.. && pwd); C=$R/rplidar_ros2_driver_amd/csrc; O=$R/build/obj
TAG=$1; shift
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I$R/include -I$C"
/opt/rocm/bin/hipcc $FL "$@" -c $C/rpl_laserscan.hip -o $O/rpl_laserscan_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/rplidar_ros2_driver_amd/lib/librplgpu_$TAG.so $O/rpl_kernels.o $O/rpl_voxel.o $O/rpl_ror.o $O/rpl_decode.o $O/rpl_msg.o $O/rpl_fuse.o $O/rpl_project.o $O/rpl_comm.o $O/rplgpu_api.o $O/rpl_laserscan_$TAG.o -ldl
echo built librplgpu_$TAG.so
