cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/dense_nw.txt
: > $O
for nw in ${NWS:-12 10 14 8}; do
  bash tools/variant_build.sh dnw$nw nufft_dense3.hip "-DNUFFT_DENSE_NW=$nw" > /dev/null 2>&1 || { echo "build $nw failed" | tee -a $O; continue; }
  for rep in 1 2; do
    echo "NW=$nw: $(NUFFT_PKG=/tmp/variants/dnw$nw python tools/stage_times.py type_1 256,256,256 1e8 1e-4 "" --one-call 2>&1 | tail -1)" | tee -a $O
  done
  echo "NW=$nw: $(NUFFT_PKG=/tmp/variants/dnw$nw python tools/stage_times.py type_1 256,256,256 3e7 1e-4 "" --one-call 2>&1 | tail -1)" | tee -a $O
done
