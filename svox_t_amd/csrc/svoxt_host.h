// svoxt_host.h -- host-side helpers shared by the translation units of
// libsvoxt_hip.so (not part of the public C ABI).
#pragma once

#include "../../include/svoxt.h"
#include "svoxt_device.h"

namespace svoxt {

// Records the text svoxt_last_error() returns on this thread and hands back `code`.
int set_error(int code, const char* fmt, const char* a = "", const char* b = "");

// hipGetLastError() -> SVOXT_OK / SVOXT_ERR_HIP (+ error text)
int check_launch(const char* what);

// Argument validation shared by every entry point (the TORCH_CHECKs of
// data_spec.hpp:57-64, 85-110 for raw pointers); `fn` names the caller in the error text.
int check_tree(const svoxt_tree* t, const char* fn);
int check_rays(const svoxt_rays* r, const char* fn);
int check_opts(const svoxt_options* o, const svoxt_tree* t, const char* fn, bool needs_basis);

// C structs -> what the kernels take by value
TreeDev to_dev(const svoxt_tree* t);
// t (may be NULL) decides the walk order of an image's tiles -- unless the lists (may be NULL) name the walk they were recorded with
RaysDev to_dev(const svoxt_rays* r, const svoxt_tree* t, const svoxt_sample_lists* l = nullptr);
Opts to_dev(const svoxt_options* o);

}  // namespace svoxt
