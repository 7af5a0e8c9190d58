#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...] -- builds anonymous-credit-tokens_amd/libact_<name>.so (same sources, extra -D flags)
# for same-box A/B runs (tools/ab_bench.sh <name>...).  Objects go to build/<name>/ (git-ignored).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/anonymous-credit-tokens_amd/csrc
out=$root/build/$name
mkdir -p "$out"
make -C "$src" host_hash.o host_pool.o node.o >/dev/null
pids=()
# FILES="k_spend_verify ..." restricts the extra flags to those translation units (default: all)
for f in engine k_misc k_spend_verify k_spend_bits k_sign k_prove k_client; do
  extra=("$@")
  if [ -n "$FILES" ] && ! [[ " $FILES " == *" $f "* ]]; then extra=(); fi
  # the range kernel's unit is built pair-interleaved in the product (csrc/Makefile BITSFLAGS); BITS= overrides that for A/B ("BITS=" = per-column form)
  if [ "$f" = k_spend_bits ]; then extra+=(${BITS--DACT_FE_PAIR_ASM}); fi
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -Wno-unused-value -DACT_CT_SECRET_TABLES "${extra[@]}" -c "$src/$f.hip" -o "$out/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
extra_libs=()
[[ " $* " == *" -DACT_ROCTX "* ]] && extra_libs=(-L/opt/rocm/lib -lrocprofiler-sdk-roctx)      # named ranges for rocprofv3 --marker-trace
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/anonymous-credit-tokens_amd/libact_$name.so" "$out"/*.o "$src/host_hash.o" "$src/host_pool.o" "$src/node.o" -lpthread "${extra_libs[@]}"
echo "built libact_$name.so"
