// Error reporting and version of the C-ABI.
#include "common.h"
#include "dvae_hip.h"
#include <atomic>

namespace dvae {
char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}
static std::atomic<long long> g_launches{0};
void count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
}  // namespace dvae

extern "C" int64_t dvae_debug_launch_count(void) { return (int64_t)dvae::g_launches.load(std::memory_order_relaxed); }
extern "C" const char* dvae_last_error(void) { return dvae::err_buf(); }
extern "C" int dvae_version(void) { return 100; }
