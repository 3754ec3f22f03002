#!/bin/bash
# Runs ON THE GPU BOX: dense-transform micro-benchmark (sizes / input layouts).
for n in ${SIZES:-34816 273744}; do for f in ${FLAGS:-0 1}; do
  timeout -k 5 60 profiles/tools/_bin/gemm_bench $n $f | head -2 | tr '\n' ' '; echo
done; done
