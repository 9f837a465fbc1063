// Error plumbing + ABI version for libnovic_hip.so (see include/novic_hip.h).
#include <string.h>
#include "novic_hip.h"

static thread_local char g_err[512] = "";

extern "C" void novic_set_error(const char* msg) {
	strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
	g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* novic_last_error(void) { return g_err; }
extern "C" int novic_abi_version(void) { return NOVIC_ABI_VERSION; }
