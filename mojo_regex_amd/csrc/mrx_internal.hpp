// Internal glue between the translation units of libmrx_hip.so (not part of any ABI).
#pragma once
#include <string>

namespace mrx {
// records the calling thread's last error message (mrx_last_error()) and returns `code`
int internal_fail(int code, const std::string& msg);
}  // namespace mrx
