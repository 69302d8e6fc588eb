// svoxt_host.h -- host-side helpers shared by the translation units of
// libsvoxt_hip.so (not part of the public C ABI).
#pragma once

namespace svoxt {

// Records the text svoxt_last_error() returns on this thread and hands back `code`.
int set_error(int code, const char* fmt, const char* a = "", const char* b = "");

// hipGetLastError() -> SVOXT_OK / SVOXT_ERR_HIP (+ error text)
int check_launch(const char* what);

}  // namespace svoxt
