#!/bin/bash
# Builds a copy of the package whose libnufft_hip.so has ONE translation unit recompiled with extra flags (A/B of
# compile-time variants in one gpurun call): tools/variant_build.sh NAME FILE.hip "-DFLAG ..." -> /tmp/variants/NAME
# (use: NUFFT_PKG=/tmp/variants/NAME python tools/ab_stack.py ...)
cd ${GRAFT_REPO_ROOT:-/root/repo}
S=/tmp/variants/$1; rm -rf $S; mkdir -p $S/obj
cp -r tensorflow-nufft_amd/tensorflow_nufft $S/
C=tensorflow-nufft_amd/csrc
FL="-O3 -std=c++17 -fPIC -Iinclude -I$C --offload-arch=gfx950 -munsafe-fp-atomics"
base=$(basename $2 .hip); base=$(basename $base .cpp)   # (.hip or .cpp translation unit)
# every variant is an EXPERIMENT build (csrc/nufft_experiment.h refuses the macros otherwise) and says so through
# nufft_hip_build_info(): the build-info unit is recompiled with the same flags
X="-DNUFFT_EXPERIMENT_BUILD $3"
/opt/rocm/bin/hipcc $FL $X -c $C/$2 -o $S/obj/$base.o || exit 1
/opt/rocm/bin/hipcc $FL $X -DNUFFT_SOURCE_DIGEST=variant_$1 -x c++ -c $C/nufft_build_info.cpp -o $S/obj/nufft_build_info.o || exit 1
OBJS="$S/obj/nufft_build_info.o"
for o in nufft_kernels nufft_dense3 nufft_wide nufft_line nufft_fft nufft_plan nufft_op; do
  if [ $o == $base ]; then OBJS="$OBJS $S/obj/$base.o"; else OBJS="$OBJS $C/_obj/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $S/tensorflow_nufft/libnufft_hip.so $OBJS -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib || exit 1
echo "built $S"
