// rt_internal.h -- shared by the host-side translation units of librt_hip.so (not part of the ABI).
#pragma once
#include <string>

namespace rt {
// Records the calling thread's error text (what rt_last_error() returns) and hands back `code`.
int set_error(const std::string &message, int code);
}
