// Library-wide pieces of the C ABI: version and the thread-local error string.
#include "common.h"

#include <cstring>

namespace ilx {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ilx

extern "C" int ilx_version(void) { return ILX_ABI_VERSION; }
extern "C" const char* ilx_last_error(void) { return ilx::error_buffer(); }
