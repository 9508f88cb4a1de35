// Error reporting shared by every entry point of libsid.so.
#include "sid_common.h"

namespace sid {

char* error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace sid

extern "C" const char* sid_last_error(void) { return sid::error_buffer(); }
extern "C" int sid_version(void) { return SID_VERSION; }
