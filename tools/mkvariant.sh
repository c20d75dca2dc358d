#!/bin/bash
# builds a variant of libfqzhip.so into ab_build_<name>/ with extra -D flags: tools/mkvariant.sh name "-DFOO=1 -DBAR=2"
set -e
NAME=$1; FLAGS=$2
OUT=/root/repo/ab_build_$NAME
mkdir -p $OUT
cd /root/repo/fastqpacker_amd/csrc
for f in fqz_api fqz_encode fqz_decode fqz_stream; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I/root/repo/include --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off $FLAGS -c $f.hip -o $OUT/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libfqzhip.so $OUT/fqz_api.o $OUT/fqz_encode.o $OUT/fqz_decode.o $OUT/fqz_stream.o
rm -f $OUT/*.o
ls -la $OUT/libfqzhip.so
