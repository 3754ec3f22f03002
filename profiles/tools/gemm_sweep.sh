#!/bin/bash
# Runs ON THE GPU BOX: dense-transform micro-benchmark (sizes / input layouts: flags 1 = no edge-less-row mask,
# 2 = last-layer inputs in separate dense [N,64] tables).
for n in ${SIZES:-273744}; do for f in ${FLAGS:-0 1 2 3}; do
  timeout -k 5 60 profiles/tools/_bin/gemm_bench $n $f | head -1
done; done
