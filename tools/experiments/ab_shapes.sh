#!/bin/bash
# pipeline at given shapes for several library builds: ab_shapes.sh "N:H N:H" lib1 lib2 ...
SH=$1; shift
for lib in "$@"; do
  echo "== $lib"
  VBX_LIB_PATH=$PWD/$lib python3 tools/experiments/pipeline_shapes.py 1 $SH | cut -c1-120
done
