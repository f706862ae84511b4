#!/bin/bash
# Builds libscpair_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../libscpair_hip.so"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$ROOT/include -I$HERE -Wno-unused-result $EXTRA_FLAGS"
OBJ="$HERE/build"
mkdir -p "$OBJ"
pids=()
for f in capi gemm wgemm spmm_bf16 tiles norm_act enc_tail adamw elementwise raw_rows eval_metrics decoder decoder_ber_ns1 decoder_ber_ns3 decoder_nb_ns1 decoder_nb_ns3; do
  if [ "$HERE/$f.hip" -nt "$OBJ/$f.o" ] || [ "$HERE/common.h" -nt "$OBJ/$f.o" ] || [ "$HERE/decoder_kernel.h" -nt "$OBJ/$f.o" ] || [ "$ROOT/include/scpair_hip.h" -nt "$OBJ/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
rm -f "$OBJ/libgemm.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o
echo "built $OUT"
# benchmark-only comparison library (rocBLAS bf16 GEMM beside the hand-written wide GEMM): not part of the product
CMP="$ROOT/benchmarks/rocblas_cmp"
if [ "$CMP/rocblas_cmp.hip" -nt "$CMP/librocblas_cmp.so" ]; then
  $HIPCC --offload-arch=gfx950 -O2 -fPIC -shared -o "$CMP/librocblas_cmp.so" "$CMP/rocblas_cmp.hip" -L/opt/rocm/lib -lrocblas -Wl,-rpath,/opt/rocm/lib
  echo "built $CMP/librocblas_cmp.so"
fi
