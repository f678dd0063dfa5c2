// host_sums.h -- CRC-32 / Adler-32 over host memory for the stream ABI (libz_mi355.so): the check value of an inflate()
// stream follows the bytes handed to the caller (zlib-rs/src/inflate/window.rs:95-168), and the exported crc32() / adler32()
// utilities work on caller memory (zlib-rs/src/crc32.rs:19-29, adler32.rs:19-87).  host_sums.cpp: carry-less-multiply folding
// and SSSE3 sums where the CPU has them (the reference dispatches the same way, crc32/pclmulqdq.rs, adler32/avx2.rs), the
// table / scalar loops otherwise.  Host code only; the batch paths compute their checksums on the GPU (checksum.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>
uint32_t zmi_host_crc32(uint32_t crc, const uint8_t* buf, size_t len);
uint32_t zmi_host_adler32(uint32_t adler, const uint8_t* buf, size_t len);
