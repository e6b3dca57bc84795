# usage: bash tools/sweeps/build_alt_src.sh SRC NAME -DFLAG=... -> scratch/alt/NAME.so: the library with csrc/SRC.hip recompiled
# under the extra flags (other objects reused); select it with JTSM_HIP_LIB=scratch/alt/NAME.so
set -e
SRC=$1; NAME=$2; shift; shift
mkdir -p scratch/alt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-gpu-rdc "$@" \
  -c jtsm_amd/csrc/$SRC.hip -o scratch/alt/$NAME.o
OBJS=$(ls jtsm_amd/lib/obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/alt/$NAME.so scratch/alt/$NAME.o $OBJS
ls -la scratch/alt/$NAME.so
