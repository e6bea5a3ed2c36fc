#include <cstdarg>
#include <cstdio>
namespace ifhip {   // the two functions of api.cpp the stage files call
static thread_local char g_msg[512];
int fail(int status, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_msg, sizeof g_msg, fmt, ap); va_end(ap); return status; }
const char* last_error() { return g_msg; }
}
// development switches: none set in these runs (the product's registry lives in weights.cpp, which the parser harness does not link)
namespace ifhip { const char* debug_switch(const char*) { return nullptr; } }
