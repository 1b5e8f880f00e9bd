// The one symbol csrc/vbx_host.cpp takes from the device side of the library, for its host-only sanitizer build
// (vox_box.rs_amd/Makefile: lib/libvbx_host_asan.so).
#include <string>

struct vbx_ctx;
static thread_local std::string g_err;
extern "C" int vbx_internal_fail(vbx_ctx *, int code, const char *msg) { g_err = msg ? msg : ""; return code; }
extern "C" const char *vbx_last_error(const vbx_ctx *) { return g_err.c_str(); }
