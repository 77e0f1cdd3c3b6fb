#!/bin/bash
# Traffic lab: where do the beyond-L2 bytes of the convolution layers come from?  Builds lab variants of the GEMM
# translation unit (-DXV_TRAFFIC_LAB=bits, csrc/gemm_bf16x3.hip) next to the product objects and, on the GPU box,
# collects FETCH_SIZE and the kernel durations of each.  usage:  bash tools/traffic_lab.sh build   (here)
#                                                               bash tools/traffic_lab.sh run     (through gpurun)
set -e
cd "$(dirname "$0")/.."
PKG=tf-kaldi-speaker_amd
VARIANTS="${VARIANTS:-0 1 2 4 8 16}"
if [ "$1" = build ]; then
  [ -n "$SKIP_PRODUCT" ] || python -c "import __graft_entry__ as g; g.build()"          # product objects are up to date
  for v in $VARIANTS; do
    d=$PKG/build/tlab_$v; mkdir -p $d
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I include -DXV_TRAFFIC_LAB=$v -c $PKG/csrc/gemm_bf16x3.hip -o $d/gemm_bf16x3.o &
  done
  wait
  for v in $VARIANTS; do
    d=$PKG/build/tlab_$v
    objs=$(ls $PKG/build/*.o | grep -v gemm_bf16x3.o)
    hipcc --offload-arch=gfx950 -shared -fPIC $objs $d/gemm_bf16x3.o -o $d/libxvec_hip.so
  done
  ls -la $PKG/build/tlab_*/libxvec_hip.so
  exit 0
fi
out=gpurun_out/tlab; mkdir -p $out
export TMPDIR=/tmp
for v in $VARIANTS; do
  echo "variant $v $(date +%T)"
  XVEC_TLAB_LIB=$PWD/$PKG/build/tlab_$v/libxvec_hip.so timeout -k 10 200 python3 tools/traffic_lab.py > $out/time_$v.txt 2>&1
  XVEC_TLAB_LIB=$PWD/$PKG/build/tlab_$v/libxvec_hip.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_$v -- python3 tools/traffic_lab.py 3 > $out/pmc_$v.log 2>&1
  python profiles/summarize_pmc.py $(find $out/pmc_$v -name "*counter_collection.csv" | head -1) > $out/fetch_$v.txt 2>&1
  rm -rf $out/pmc_$v
  grep -A1 "w14p2_kernel\|w1p3_kernel" $out/fetch_$v.txt | grep -v "^--" | paste - - | sed 's/  */ /g' > $out/fetch_${v}_short.txt
  cat $out/time_$v.txt | tail -3
done
