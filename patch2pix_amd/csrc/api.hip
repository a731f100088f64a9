// Library-wide pieces of the C ABI: version and thread-local error string.
#include "p2p_common.h"

namespace p2p {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace p2p

#ifdef P2P_EXPERIMENT
extern "C" int p2p_version(void) { return 102 | P2P_VERSION_EXPERIMENT; }
#else
extern "C" int p2p_version(void) { return 102; }
#endif
extern "C" const char *p2p_last_error(void) { return p2p::g_err; }
