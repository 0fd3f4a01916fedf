// Error reporting + version of the C ABI (include/transfuser_hip.h).
#include "tf_common.h"
#include "../../include/transfuser_hip.h"
#include <stdarg.h>

namespace tf {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace tf

extern "C" int tf_version(void) { return 100; }
extern "C" const char* tf_last_error(void) { return tf::g_err; }
