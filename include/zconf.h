/* zconf.h -- drop-in name (libz-rs-sys/include/zconf.h): the configuration types and macros (z_const, z_off_t,
 * z_off64_t, z_size_t, ZEXTERN / ZEXPORT, OF, MAX_WBITS ...) are defined in zmi355_zlib.h. */
#ifndef ZCONF_H
#define ZCONF_H
#include "zmi355_zlib.h"
#endif
