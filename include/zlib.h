/* zlib.h -- drop-in name for programs that `#include "zlib.h"` (libz-rs-sys-cdylib/example.c:9, zpipe.c:18):
 * the declarations live in zmi355_zlib.h, the library is zlib_rs_amd/libz_mi355.so. */
#ifndef ZLIB_H
#define ZLIB_H
#include "zmi355_zlib.h"
#endif
