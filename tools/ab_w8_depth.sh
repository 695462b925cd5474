#!/bin/bash
# A/B: 3-D float w = 8 tiles of depth 8 (default) against depth 4 (NUFFT_HIP_W8_DEPTH4=1)
for args in "8e5 128,128,128 type_1" "8e5 128,128,128 type_2" "1e7 128,128,128 type_1" "1e7 128,128,128 type_2" "1e8 256,256,256 type_1" "1e8 256,256,256 type_2"; do
  echo "--- depth 8"; python tools/time_case3.py $args
  echo "--- depth 4"; NUFFT_HIP_W8_DEPTH4=1 python tools/time_case3.py $args
done
