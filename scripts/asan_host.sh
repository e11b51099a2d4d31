#!/bin/bash
# usage (build container, no GPU): bash scripts/asan_host.sh
# AddressSanitizer + UBSan over the host side of libpgx (file parsers, image builders, index / tag writers): the three C++
# sources (+ the GBZ reader) are compiled without HIP into /tmp/libpgx_asan.so and the parser fuzz + format tests run against it.
set -e
cd "$(dirname "$0")/.."
cat > /tmp/pgx_asan_stub.cpp <<'CPP'
struct pgx_index;
void pgx_release_device_images(pgx_index *) {}
// the device entry points live in pgx_runtime.hip: "no device" stubs so that the ctypes binding finds every symbol
#define STUB(name) extern "C" int name() { return 4; /* PGX_ERR_NO_DEVICE */ }
STUB(pgx_index_to_device) STUB(pgx_device_count) STUB(pgx_device_name) STUB(pgx_rank_batch) STUB(pgx_extend_batch) STUB(pgx_count_batch)
STUB(pgx_tag_query_batch) STUB(pgx_locate_batch) STUB(pgx_locate_next_batch) STUB(pgx_decompress_sa) STUB(pgx_batch_create)
STUB(pgx_batch_upload) STUB(pgx_batch_run) STUB(pgx_batch_result) STUB(pgx_batch_counts) STUB(pgx_batch_timing) STUB(pgx_batch_free)
STUB(pgx_find_mems_batch) STUB(pgx_merge_tags) STUB(pgx_batch_device_result) STUB(pgx_batch_spec_stats) STUB(pgx_comm_free) STUB(pgx_comm_init)
STUB(pgx_comm_unique_id) STUB(pgx_exchange_download) STUB(pgx_exchange_mems) STUB(pgx_find_mems_function_batch) STUB(pgx_find_mems_sharded)
STUB(pgx_lf_batch) STUB(pgx_merge_tags_gbz)
CPP
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined \
    -o /tmp/libpgx_asan.so pangenome-index_amd/csrc/pgx_index.cpp pangenome-index_amd/csrc/pgx_sdsl.cpp pangenome-index_amd/csrc/pgx_build.cpp pangenome-index_amd/csrc/pgx_gbz.cpp /tmp/pgx_asan_stub.cpp
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1 LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) \
    PGX_LIB=/tmp/libpgx_asan.so python -m pytest tests/test_fuzz_parsers.py tests/test_formats.py tests/test_image.py tests/test_locate.py tests/test_pairs_image.py tests/test_gbz.py -x -q -m "not gpu" -p no:cacheprovider
