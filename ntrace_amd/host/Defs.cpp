// Defs.cpp -- error model of src/framework/base/Defs.hpp:142-151 / Defs.cpp:257-365.
#include "Defs.hpp"

#include <cstdarg>
#include <cstdio>

namespace FW {

static thread_local String s_error;
static thread_local bool s_hasError = false;

void fail(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    fprintf(stderr, "\nFW::fail: %s\n", buf);
    throw FatalError{buf};
}

void setError(const char* fmt, ...)
{
    if (s_hasError) return;  // the first error sticks (Defs.cpp:308)
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    s_error = buf;
    s_hasError = true;
}

bool hasError(void) { return s_hasError; }
const String& getError(void) { return s_error; }
void clearError(void) { s_hasError = false; s_error.clear(); }
void failIfError(void) { if (s_hasError) fail("%s", s_error.c_str()); }

}  // namespace FW
