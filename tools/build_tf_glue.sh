#!/bin/bash
# One command for whoever has TensorFlow-ROCm: builds the drop-in `_nufft_ops.so` (the reference loads a library of
# that name, tensorflow_nufft/python/ops/nufft_ops.py:26-27) from csrc/tf_glue/nufft_tf_ops.cc against the installed
# TensorFlow and the in-tree libnufft_hip.so, then runs tests/test_tf_dropin.py on it.
#   tools/build_tf_glue.sh [OUT_DIR]      (default: tensorflow-nufft_amd/tf_dropin)
# TensorFlow is NOT installed in this repository's image: nothing here has been run against the real headers; the
# glue is type-checked and its shape function executed against tests/tf_api_stub/ instead.
set -e
cd "$(dirname "$0")/.."
OUT=${1:-tensorflow-nufft_amd/tf_dropin}
python3 -c 'import tensorflow' 2>/dev/null || { echo "TensorFlow is not importable here: nothing to build against" >&2; exit 3; }
make -C tensorflow-nufft_amd/csrc -j8
TF_CFLAGS=$(python3 -c 'import tensorflow as tf; print(" ".join(tf.sysconfig.get_compile_flags()))')
TF_LFLAGS=$(python3 -c 'import tensorflow as tf; print(" ".join(tf.sysconfig.get_link_flags()))')
mkdir -p "$OUT"
cp tensorflow-nufft_amd/tensorflow_nufft/libnufft_hip.so "$OUT/"
${HIPCC:-/opt/rocm/bin/hipcc} -std=c++17 -O2 -shared -fPIC tensorflow-nufft_amd/csrc/tf_glue/nufft_tf_ops.cc \
  -o "$OUT/_nufft_ops.so" $TF_CFLAGS -Iinclude -L"$OUT" -lnufft_hip -Wl,-rpath,'$ORIGIN' $TF_LFLAGS
echo "built $OUT/_nufft_ops.so"
NUFFT_TF_OPS_SO="$PWD/$OUT/_nufft_ops.so" python3 -m pytest tests/test_tf_dropin.py -q -m gpu
