#!/bin/bash
# Development aid: build a variant of libtdc_gpu.so in which ONE source file is compiled with extra flags.
#   tools/build_variant.sh NAME file.hip "-DFOO -DBAR=1"   -> tudocomp_amd/lib/variants/NAME.so   (run with TDC_GPU_LIB=that file)
set -e
NAME=$1; SRC=$2; FLAGS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/tudocomp_amd/csrc
make -s
mkdir -p ../lib/variants
STEM=${SRC%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $FLAGS -c $SRC -o ../lib/variants/${NAME}_$STEM.o
OBJS=$(ls ../lib/obj/*.o | grep -v "/$STEM.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib/variants/$NAME.so $OBJS ../lib/variants/${NAME}_$STEM.o
rm -f ../lib/variants/${NAME}_$STEM.o
echo built $R/tudocomp_amd/lib/variants/$NAME.so
