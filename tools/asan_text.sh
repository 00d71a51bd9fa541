#!/bin/bash
# The VCF text writer (gtx_vcf.cpp writes a record's sample columns through a raw pointer into room made beforehand) under
# AddressSanitizer: its object alone is rebuilt with -fsanitize=address, linked with the other objects into libgtx_asan.so, and
# the CPU tests that make text run against it.  Usage: bash tools/asan_text.sh   (from the repository's root, after a build)
set -eu
cd "$(dirname "$0")/../graphtyper_amd/csrc"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
rm -rf build_asan && cp -r build build_asan
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -shared-libasan -fno-omit-frame-pointer -c -o build_asan/gtx_vcf.o gtx_vcf.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libasan -o ../libgtx_asan.so build_asan/*.o -lz -ldl
cd ../..
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 LD_PRELOAD=$RT GTX_LIB=libgtx_asan.so python -m pytest tests/test_vcf_text.py tests/test_sv_vcf.py tests/test_emu_parity.py tests/test_bam_ingest.py -x -q -m "not gpu"
rm -rf graphtyper_amd/csrc/build_asan graphtyper_amd/libgtx_asan.so
